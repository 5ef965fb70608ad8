python tools/probes/wk_probe.py > gpurun_out/wk_probe_plane.txt 2>&1
python tools/probes/wk_s1_probe.py > gpurun_out/wk_s1_probe_plane.txt 2>&1
for cfg in c2 s1 c4; do
for rep in 1 2; do
for opt in 0x1 0x201; do
  r=$(timeout 300 python tools/probes/bench_wkopt.py $opt --config $cfg --headline-only --cpu-steps 0 --steps 30 --warmup 10 2>/dev/null | python -c "import sys,json; print(round(json.loads(sys.stdin.readlines()[-1])['ms_per_step'],4))")
  echo "RESULT $cfg opt=$opt $r"
done; done; done
timeout 600 python -m pytest tests/test_gpu_ops.py -x -q 2>&1 | tail -2
