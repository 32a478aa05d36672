# A/B of the timed region: schema 5 (first timed key batch encoded during the warm-up) vs schema 6 (self-contained), --scale-only
mkdir -p gpurun_out/c15 && cd $GRAFT_REPO_ROOT
O=gpurun_out/c15
for rep in 1 2 3; do
  for v in bench_prev_tmp.py bench.py; do
    python $v --scale-only --steps 20 --warmup 5 2>>$O/err.log | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', 20, 5, round(d['value'],1), d['ms_per_step'])" >> $O/ab.txt
    python $v --scale-only --steps 200 --warmup 10 2>>$O/err.log | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', 200, 10, round(d['value'],1), d['ms_per_step'])" >> $O/ab.txt
  done
done
python bench.py --scale-only --steps 22 --warmup 8 2>>$O/err.log | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench.py', 22, 8, round(d['value'],1), d['ms_per_step'])" >> $O/ab.txt
cat $O/ab.txt; tail -5 $O/err.log
