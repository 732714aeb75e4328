set -e
for fam in blob knot; do
for kv in "RTX_W_SHADE 3,4,5,6" "RTX_W_LEAF 3,4,5" "RTX_ALSO_SHADE 16,24,32" "RTX_ALSO_LEAF 12,16,24"; do
  set -- $kv
  timeout -k 10 120 python tools/knob_probe.py $1 $2 dragon $fam 2>&1 | grep pruned
done
done
