for lib in "" $PWD/ab/libgraft_base.so; do for wl in default single; do GRAFT_LIB=$lib timeout -k 10 100 python bench.py --workload $wl --no-cpu --no-frame512 --steps 20 --warmup 3 2>/dev/null | python -c "
import json,sys
b=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$wl', '${lib:-in-tree}', round(b['roofline']['kernel_ms'],4), 'ms', b['rays_per_frame'])"; done; done
