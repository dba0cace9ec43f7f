# on the GPU box: f32x3 conv layers, register-staged split vs plane kernel (same box)   bash tools/x3_ab.sh [SIZE BATCH]
SIZE=${1:-512}; BATCH=${2:-32}
for p in ${3:-0 1 2}; do
  echo "== f32x3 planes=$p"
  timeout -k 10 300 python tools/bench_ops.py --size $SIZE --batch $BATCH --bf16 2 --x3planes $p --layers 1,2,3,4,5,6,7 2>&1 | grep -E "conv s2|totals" || exit 1
done
