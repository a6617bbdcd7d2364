E="GLR_FORCE_DIST=1 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1"
b() { name=$1; gb=$2; steps=$3; shift 3; env "$@" timeout -k 10 400 python bench.py --no-cpu-baseline --global-batch $gb --steps $steps 2>>gpurun_out/r03u.err | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); c=d['config']; print('$name:', round(d['ms_per_step'],2), 'ms/step')"; }
b "b32 dist hook-free" 32 40 $E MASTER_PORT=29701 GLR_REDUCER_OVERLAP=0
b "b32 dist hook-driven" 32 40 $E MASTER_PORT=29702 GLR_REDUCER_OVERLAP=1
b "b32 dist hook-free" 32 40 $E MASTER_PORT=29703 GLR_REDUCER_OVERLAP=0
b "b32 dist hook-driven" 32 40 $E MASTER_PORT=29704 GLR_REDUCER_OVERLAP=1
b "b128 dist hook-free" 128 15 $E MASTER_PORT=29705 GLR_REDUCER_OVERLAP=0
b "b128 dist hook-driven" 128 15 $E MASTER_PORT=29706 GLR_REDUCER_OVERLAP=1
