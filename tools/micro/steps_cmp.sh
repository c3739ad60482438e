for s in "20 5" "20 5" "200 20" "20 5"; do
  set -- $s
  python3 bench.py --steps $1 --warmup $2 --tune-cache /tmp/tt.txt --no-cpu-baseline --no-secondary 2>/dev/null | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['steps'], j['warmup'], round(j['ms_per_step'],4), [round(x,4) for x in j['config']['repeat_ms_per_step']['runs']])"
done
