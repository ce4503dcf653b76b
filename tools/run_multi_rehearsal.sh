set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
# two gloo ranks on the one GPU of the box: the multi-rank path of bench.py at reduced sizes (row blocks + gathers through
# the host, element ranges, the strong-scaled record of configuration 5 at 4096^2 / 2^24)
timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 2 --steps 2 --warmup 1 --size 1024 --bitwise-log2n 20 --dist-backend gloo --no-cpu --config5-size 4096 --config5-log2n 24 > gpurun_out/r3_multi_gloo.json 2> gpurun_out/r3_multi_gloo.err
echo gloo2 rc=$?
# one RCCL rank under torch.distributed.run: the collectives of the real path with a single member
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29612 bench.py --gpus 1 --steps 2 --warmup 1 --no-cpu --bitwise-log2n 22 --no-extras > gpurun_out/r3_multi_nccl1.json 2> gpurun_out/r3_multi_nccl1.err
echo nccl1 rc=$?
