for p in main sketch; do for a in "--sketch-groups 4 --steps 12 --lanes 12" "--sketch-groups 3 --steps 9 --lanes 9"; do
  v=$(MUSED_BENCH_PRIO=$p timeout -k 10 300 python bench.py --no-cpu-baseline $a 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['stages_ms']['swfd_groups'], round(d['value']), round(d['p50_window_latency_ms']))")
  echo "prio=$p $a -> $v"
done; done
