set -e
for wl in cfg4 cfg3 cfg5; do
    echo "== wl=$wl"
    timeout -k 10 120 python bench.py --workload $wl --no-cpu --no-frame512 --steps 3 --warmup 1 | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'])"
done
