set -e
B="timeout -k 10 200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-hbm-record"
for o in "" "--bvh-option bvh.sah.use_splits=1" "--bvh-option bvh.sah.use_splits=1 --bvh-option bvh.sah.extra_node_budget=1.0" "--bvh-option mipt.optimize_passes=6" "--bvh-option bvh.sah.num_bins=128" "--bvh-option bvh.sah.traversal_cost=1"; do
  echo "== $o"
  $B $o 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print(round(d['value']), 'Mrays/s', 'boxes/ray', round(d['nodes_per_ray'],2), 'tris/ray', round(d['tris_per_ray'],2), 'extend ms', round(d['kernel_ms_per_step']['extend'],2), 'build s', d['config']['bvh_build_s'], 'ties', d['tie_rays'], 'depth', d['tree_depth'], 'radiance', d['mean_radiance'])"
done
