# bash tools/probes/env_sweep.sh <config> <ENVVAR> "<values>": ms/step of bench.py per value of a probe environment variable
for v in $3; do
  r=$(env $2=$v timeout 400 python bench.py --config $1 --headline-only --cpu-steps 0 --steps 20 --warmup 5 2>/dev/null | python -c "import sys,json; print(round(json.loads(sys.stdin.readlines()[-1])['ms_per_step'],4))")
  echo "RESULT $1 $2=$v $r"
done
