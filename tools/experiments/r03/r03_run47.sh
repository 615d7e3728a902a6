cd $GRAFT_REPO_ROOT
for m in layer cell dir; do echo "== RNH_LSTM_STREAMS=$m"; RNH_LSTM_STREAMS=$m python tools/predict_bench.py 2>&1 | grep -v amdgpu.ids; done
