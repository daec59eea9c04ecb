#!/bin/bash
# extra measurements of a round: the sharded configs, launcher forms of bench.py on a 1-GPU box, Stream.run end to end
TAG=${1:-r03}
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out
for c in c4 c5; do
  timeout 600 python bench.py --config $c --steps 40 --warmup 3 > $O/${TAG}_bench_$c.json 2>$O/${TAG}_bench_$c.err; tail -c 700 $O/${TAG}_bench_$c.json; echo
done
# the driver's launcher form with ONE rank: RANK / WORLD_SIZE from the environment
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29536 \
  bench.py --gpus 1 --steps 20 --warmup 3 --cpu-windows 0 --no-cold-start --no-mode-a > $O/${TAG}_headline_torchrun1.json 2>$O/${TAG}_headline_torchrun1.err; tail -c 300 $O/${TAG}_headline_torchrun1.json; echo
# --gpus 2 on a 1-GPU box: must refuse (never a 1-GPU number under a 2-GPU label)
python bench.py --gpus 2 > $O/${TAG}_gpus2_refused.txt 2>&1; echo "rc=$? $(tail -1 $O/${TAG}_gpus2_refused.txt)"
# two ranks sharing the one GPU over gloo: the exchange step of c4 and the max-over-ranks timing
NMX_BENCH_FORCE_DEVICE=0 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29534 \
  bench.py --gpus 2 --config c4 --steps 20 --warmup 3 --backend gloo > $O/${TAG}_c4_2rank_gloo.json 2>$O/${TAG}_c4_2rank_gloo.err; tail -c 400 $O/${TAG}_c4_2rank_gloo.json; echo
NMX_BENCH_FORCE_DEVICE=0 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29535 \
  bench.py --gpus 2 --steps 20 --warmup 3 --cpu-windows 0 --backend gloo > $O/${TAG}_headline_2rank_gloo.json 2>$O/${TAG}_headline_2rank_gloo.err; tail -c 300 $O/${TAG}_headline_2rank_gloo.json; echo
# the headline as ONE 256-channel array over two ranks (strong scaling: joint re-reference, one all-reduce per step)
NMX_BENCH_FORCE_DEVICE=0 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29537 \
  bench.py --gpus 2 --scaling strong --steps 20 --warmup 3 --backend gloo > $O/${TAG}_headline_strong_2rank_gloo.json 2>$O/${TAG}_headline_strong_2rank_gloo.err; tail -c 400 $O/${TAG}_headline_strong_2rank_gloo.json; echo
python tools/bench_stream.py > $O/${TAG}_stream_end_to_end.json 2>/dev/null; cat $O/${TAG}_stream_end_to_end.json
python tools/bench_host.py > $O/${TAG}_host_boundary.json 2>/dev/null; cat $O/${TAG}_host_boundary.json
