set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 2 --steps 2 --warmup 1 --size 1024 --bitwise-log2n 20 --dist-backend gloo --no-cpu > gpurun_out/r2_multi_gloo.json 2> gpurun_out/r2_multi_gloo.err
echo gloo2 rc=$?
ls -la /tmp/mpcx_ckpt_29611 | head
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29612 bench.py --gpus 1 --steps 2 --warmup 1 --no-cpu --bitwise-log2n 22 > gpurun_out/r2_multi_nccl1.json 2> gpurun_out/r2_multi_nccl1.err
echo nccl1 rc=$?
