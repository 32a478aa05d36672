run() { echo "== $1"; shift; env "$@" XMEM_BENCH_PARITY_TRACE=1 timeout 300 python bench.py --no-kernel-trace --no-extra-modes --plain-steps 0 --cpu-frames 4 --steps 40 2>&1 >/dev/null | grep parity | cut -c1-100; }
run "V0 default" A=1
echo "== probe2 (GPU only: early vs no early on a fresh net)"; timeout 300 python tools/probes/early_readout_probe2.py 2>&1 | tail -1
run "V4 early only inside a batch" XMEM_EARLY_SAME_GROUP=1
