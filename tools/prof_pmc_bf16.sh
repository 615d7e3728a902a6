# PMC counters of the bf16-storage kernels (tools/kbench_bf16.py), one rocprofv3 --pmc pass per counter group.
#   usage (GPU box): bash tools/prof_pmc_bf16.sh <tag> [kbench filter]   -> gpurun_out/pmc_<tag>/summary.json
set +e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
tag=${1:-r02bf}
flt=${2:-lstm}
out=gpurun_out/pmc_$tag
mkdir -p $out
i=0
for grp in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES SQ_INSTS_MFMA SQ_WAIT_ANY GRBM_GUI_ACTIVE" \
           "SQ_ACTIVE_INST_ANY SQ_WAVES SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS" \
           "FETCH_SIZE" "WRITE_SIZE"; do
    i=$((i + 1))
    rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $out/pass$i -- python tools/kbench_bf16.py $flt > $out/pass$i.log 2> $out/pass$i.err || { echo "pass $i FAILED"; tail -5 $out/pass$i.err; }
done
python tools/pmc_summary.py $out --json $out/summary.json --hbm-bf16 $out/lstm_bf16_kernel_hbm_bytes.json
find $out -name '*kernel_trace.csv' -delete
find $out -name '*counter_collection.csv' -delete
