#!/bin/bash
# For a node with several MI355X (the builder's boxes have one): the slab path's two knobs, measured.
#   tools/scaling_sweep.sh [NX] [GPU counts...]        default: 4096, 1 2 4 8
# For every GPU count: row chunks per exchange (--chunks 1|2|4) x CUs kept out of the persistent row kernels for RCCL's
# send/recv kernels (NIWQG_AMD_SLAB_RESERVE_CUS 0|16|32).  One JSON line per run in scaling_sweep.out; the line carries
# exchange_ms_per_step and host_dispatches_per_step next to the value.  DESIGN.md section 9 says what to expect.
set -u
NX=${1:-4096}
shift || true
GPUS=${*:-"1 2 4 8"}
OUT=scaling_sweep.out
: > $OUT
for n in $GPUS; do
  if [ "$n" = 1 ]; then
    echo "== 1 GPU" | tee -a $OUT
    python bench.py --gpus 1 --nx $NX --steps 50 --warmup 10 --no-cpu-baseline 2>/dev/null | grep '^{' | tee -a $OUT
    continue
  fi
  for chunks in 1 2 4; do
    for reserve in 0 16 32; do
      echo "== $n GPUs, --chunks $chunks, NIWQG_AMD_SLAB_RESERVE_CUS=$reserve" | tee -a $OUT
      NIWQG_AMD_SLAB_RESERVE_CUS=$reserve timeout -k 10 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n \
        --master-addr 127.0.0.1 --master-port $((29500 + n * 10 + chunks)) bench.py --gpus $n --nx $NX --steps 50 --warmup 10 \
        --chunks $chunks 2>/dev/null | grep '^{' | tee -a $OUT
    done
  done
  # the other link: ONE process, rank r on device r, blocks cross by peer copies (SDMA over xGMI, no CUs) -- the A/B against RCCL's
  # send/recv kernels, which compete for CUs with the persistent row kernels
  for chunks in 1 2 4; do
    echo "== $n GPUs, --link peers, --chunks $chunks" | tee -a $OUT
    timeout -k 10 900 python bench.py --gpus $n --link peers --nx $NX --steps 50 --warmup 10 --chunks $chunks 2>/dev/null | grep '^{' | tee -a $OUT
  done
  # ... and single-chunk exchanges kept on the exchange stream (they run on the compute stream by default)
  echo "== $n GPUs, --chunks 1, NIWQG_AMD_SLAB_INLINE=0" | tee -a $OUT
  NIWQG_AMD_SLAB_INLINE=0 timeout -k 10 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n \
    --master-addr 127.0.0.1 --master-port $((29700 + n)) bench.py --gpus $n --nx $NX --steps 50 --warmup 10 --chunks 1 2>/dev/null | grep '^{' | tee -a $OUT
done
