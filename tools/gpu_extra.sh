#!/bin/bash
# extra measurements of a round: other BASELINE configs, the sharded configs, a 2-rank run on ONE GPU
TAG=${1:-r02}
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out
timeout 600 python tools/bench_configs.py > $O/${TAG}_configs.json 2>$O/${TAG}_configs.err; cat $O/${TAG}_configs.json
for c in c4 c5; do
  timeout 600 python bench.py --config $c --steps 4 --warmup 2 > $O/${TAG}_bench_$c.json 2>$O/${TAG}_bench_$c.err; tail -c 900 $O/${TAG}_bench_$c.json
done
# two ranks on the one GPU of this box: does the RCCL ("nccl") process group initialise, does the exchange step run?
NMX_BENCH_FORCE_DEVICE=0 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 \
  bench.py --gpus 2 --config c4 --steps 3 --warmup 1 > $O/${TAG}_c4_2rank_nccl.json 2>$O/${TAG}_c4_2rank_nccl.err; tail -c 600 $O/${TAG}_c4_2rank_nccl.json; tail -5 $O/${TAG}_c4_2rank_nccl.err
NMX_BENCH_FORCE_DEVICE=0 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29534 \
  bench.py --gpus 2 --config c4 --steps 3 --warmup 1 --backend gloo > $O/${TAG}_c4_2rank_gloo.json 2>$O/${TAG}_c4_2rank_gloo.err; tail -c 600 $O/${TAG}_c4_2rank_gloo.json
NMX_BENCH_FORCE_DEVICE=0 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29535 \
  bench.py --gpus 2 --steps 3 --warmup 1 --cpu-windows 0 > $O/${TAG}_headline_2rank_nccl.json 2>$O/${TAG}_headline_2rank_nccl.err; tail -c 400 $O/${TAG}_headline_2rank_nccl.json; tail -3 $O/${TAG}_headline_2rank_nccl.err
