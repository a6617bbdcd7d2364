run() { echo "== $*"; env "$@" timeout -k 10 300 python tools/dbg_imggraph.py 2>&1 | grep "^iter\|^graph\|hipGraph replay"; }
run MODE=img DEBUG_CLR_GRAPH_PACKET_CAPTURE=1
run MODE=img
E="GLR_FORCE_DIST=1 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1"
for ov in 0 1; do for st in 0 1; do
echo "== DP overlap=$ov streams=$st"
env $E MASTER_PORT=2960$st GLR_REDUCER_OVERLAP=$ov STREAMS=$st timeout -k 10 300 python tools/dbg_txtgraph.py 2>&1 | grep "^step\|clip_state\|hipGraph replay" | cut -c1-150
done; done
echo "== single streams=0"; STREAMS=0 timeout -k 10 300 python tools/dbg_txtgraph.py 2>&1 | grep "^step\|clip_state\|hipGraph replay" | cut -c1-150
