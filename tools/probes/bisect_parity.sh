run() { echo "== $1"; shift; env "$@" XMEM_BENCH_PARITY_TRACE=1 timeout 300 python bench.py --no-kernel-trace --no-extra-modes --plain-steps 0 --cpu-frames 4 --steps 40 2>&1 >/dev/null | grep "per-frame IoU\|again" | cut -c1-150; }
run "early on (static buffers only)" XMEM_EARLY_READOUT=1
