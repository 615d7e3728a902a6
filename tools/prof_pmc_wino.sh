# PMC evidence for the three Winograd kernels that make up ~80 % of the step (conv_wino_kernel<LSTM>, <STORE>,
# wino_wgrad_lds_kernel) at BASELINE config 2 shapes: one rocprofv3 --pmc pass per counter group (SQ: 8 slots; FETCH_SIZE
# and WRITE_SIZE need passes of their own: MI355X_MICROARCH.md "rocprofv3 PMC slots"), python directly after `--`.
#   usage (GPU box): bash tools/prof_pmc_wino.sh <tag> [kbench filter, default lstm; lstm44 = the F(4x4,3x3) cell and its input transform]
#                    -> gpurun_out/pmc_<tag>/{summary.json,lstm_kernel_hbm_bytes.json | lstm44_kernel_hbm_bytes.json}
set +e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
tag=${1:-r02}
flt=${2:-lstm}
out=gpurun_out/pmc_$tag
mkdir -p $out
i=0
for grp in "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES SQ_INSTS_MFMA" \
           "SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVES SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE" \
           "FETCH_SIZE" "WRITE_SIZE"; do
    i=$((i + 1))
    rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $out/pass$i -- python tools/kbench.py $flt > $out/pass$i.log 2> $out/pass$i.err || { echo "pass $i FAILED"; tail -5 $out/pass$i.err; }
    echo "pass $i done: $grp"
done
if [ "$flt" = lstm44 ]; then python tools/pmc_summary.py $out --json $out/summary.json --hbm44 $out/lstm44_kernel_hbm_bytes.json
else python tools/pmc_summary.py $out --json $out/summary.json --hbm $out/lstm_kernel_hbm_bytes.json; fi
find $out -name '*kernel_trace.csv' -delete
find $out -name '*counter_collection.csv' -delete
cat $out/summary.json
