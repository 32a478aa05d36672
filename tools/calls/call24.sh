# key batch 2 / 3 against 4 under the self-contained timed region (a shorter first key pass weighs less on a short stream)
mkdir -p gpurun_out/c24 && cd $GRAFT_REPO_ROOT
O=gpurun_out/c24
for rep in 1 2; do
  for kb in 4 2 3; do
    for st in "20 5" "200 10"; do
      set -- $st
      python bench.py --scale-only --key-batch $kb --steps $1 --warmup $2 2>>$O/err.log | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('kb', $kb, 'steps', $1, round(d['value'],1), round(d['ms_per_step'],4))" >> $O/kb.txt
    done
  done
done
cat $O/kb.txt
