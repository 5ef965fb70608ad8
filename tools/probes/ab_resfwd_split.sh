for cfg in c5 s1; do for rep in 1 2; do for w in 31 63; do
  r=$(timeout 400 python tools/probes/bench_attnopt.py $w --config $cfg --headline-only --cpu-steps 0 --steps 30 --warmup 10 2>/dev/null | python -c "import sys,json; print(round(json.loads(sys.stdin.readlines()[-1])['ms_per_step'],4))")
  echo "RESULT $cfg attn-word=$w $r"
done; done; done
