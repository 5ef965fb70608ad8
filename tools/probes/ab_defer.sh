for cfg in c5 c5p c2 s2; do for rep in 1 2; do
  for v in bench.py tools/probes/bench_nodefer.py; do
  r=$(timeout 400 python $v --config $cfg --headline-only --cpu-steps 0 --steps 30 --warmup 10 2>/dev/null | python -c "import sys,json; print(round(json.loads(sys.stdin.readlines()[-1])['ms_per_step'],4))")
  echo "RESULT $cfg $v $r"
done; done; done
