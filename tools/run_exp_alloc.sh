set -e
export QUADS="1" HOST=0
echo "--- default allocation"; timeout -k 10 240 python tools/exp_quad.py sphere28m
echo "--- pow2 allocation"; MIPT_ALLOC_POW2=1 timeout -k 10 240 python tools/exp_quad.py sphere28m
