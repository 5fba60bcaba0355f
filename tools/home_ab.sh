mkdir -p gpurun_out/r06
L=gpurun_out/r06/home_ab.txt
: > $L
for lib in "" nopf; do
  echo "== C4 lib=${lib:-default}" >> $L
  python tools/probe.py chain 1000000 64 200 --init true --prune 3 --sweeps 8 --timing ${lib:+--lib $lib} 2>&1 | tail -3 >> $L
done
for lib in "" nopf w3 w3nopf; do
  echo "== C3 lib=${lib:-default}" >> $L
  python tools/probe.py chain 1000000 16 100 --init true --pcrp --prune 3 --sweeps 8 --timing ${lib:+--lib $lib} 2>&1 | tail -3 >> $L
done
for lib in "" nopf; do
  echo "== C5 lib=${lib:-default}" >> $L
  python tools/probe.py chain 2000000 128 200 --init true --pcrp --prune 3 --sweeps 6 --timing ${lib:+--lib $lib} 2>&1 | tail -3 >> $L
done
cat $L
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "every_window_pruned or short_steps or benchmarked_mode or pruning_does_not or safe_stay_windows_against or order_with_repeats or uniform_exactly" --durations=8 2>&1 | tail -15 > gpurun_out/r06/home_tests.txt
cat gpurun_out/r06/home_tests.txt
