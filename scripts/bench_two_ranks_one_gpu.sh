export MASTER_ADDR=127.0.0.1 MASTER_PORT=29533 WORLD_SIZE=2 LOCAL_RANK=0 PAI_DIST_BACKEND=gloo
RANK=1 timeout -k 10 300 python bench.py --gpus 2 --steps 5 --warmup 2 --batch 16 > gpurun_out/bench2_r1.log 2>&1 &
RANK=0 timeout -k 10 300 python bench.py --gpus 2 --steps 5 --warmup 2 --batch 16 > gpurun_out/bench2_r0.log 2>&1
wait
tail -c 600 gpurun_out/bench2_r0.log; echo; tail -3 gpurun_out/bench2_r1.log
