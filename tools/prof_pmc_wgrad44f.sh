# PMC passes on the fused F(4x4)-tile weight gradient (wf12_wgrad_kernel) at BASELINE config 2's cell problem: issue / co-execution counters and the memory path.
#   usage (GPU box): bash tools/prof_pmc_wgrad44f.sh <tag>    -> gpurun_out/pmc_<tag>/summary.json
set +e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
out=gpurun_out/pmc_${1:-wf}
mkdir -p $out
i=0
for grp in "SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_INST_CYCLES_SALU SQ_INST_CYCLES_VMEM_RD SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_LDS SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU" \
           "FETCH_SIZE" "WRITE_SIZE"; do                 # (a TA_* / TCP_* group hung rocprofv3 for 7 minutes on this pool in round 6: not collected)
    i=$((i + 1))
    rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $out/pass$i -- python tools/kbench.py lstm.wgrad > $out/pass$i.log 2> $out/pass$i.err || { echo "pass $i FAILED"; tail -5 $out/pass$i.err; }
    echo "pass $i done: $grp"
done
python tools/pmc_summary.py $out --json $out/summary.json
find $out -name '*kernel_trace.csv' -delete
find $out -name '*counter_collection.csv' -delete
