#!/bin/bash
# why did MIOpen search (naive_conv_* kernels) under rocprofv3 with the find-db "active"?
# Run from the repo root on the GPU box.  Writes gpurun_out/miopen/*.
OUT=$(pwd)/gpurun_out/miopen
ROOT=$(pwd)
mkdir -p "$OUT"
echo "[probe] plain run (MIOpen log level 5, filtered)"
MIOPEN_LOG_LEVEL=5 timeout -k 10 400 python bench.py --steps 3 --warmup 2 --no-cpu-baseline > "$OUT/plain.json" 2> "$OUT/plain.err.full"
grep -i "find-db\|finddb\|FindDb\|userdb\|user db\|ufdb\|naive\|Find(\|FindSolution\|db path\|GetUserDbPath\|GetFindDbPath\|Perf Db\|ReadonlyRamDb\|\[bench\]" "$OUT/plain.err.full" | cut -c1-400 | head -400 > "$OUT/plain.err"
wc -l "$OUT/plain.err.full" >> "$OUT/plain.err"; rm -f "$OUT/plain.err.full"
echo "[probe] rocprofv3 run (same logging)"
(cd /tmp && export TMPDIR=/tmp && MIOPEN_LOG_LEVEL=5 timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof" -- python3 "$ROOT/bench.py" --steps 3 --warmup 2 --no-cpu-baseline > "$OUT/prof.json" 2> "$OUT/prof.err.full")
grep -i "find-db\|finddb\|FindDb\|userdb\|user db\|ufdb\|naive\|Find(\|FindSolution\|db path\|GetUserDbPath\|GetFindDbPath\|Perf Db\|ReadonlyRamDb\|\[bench\]" "$OUT/prof.err.full" | cut -c1-400 | head -400 > "$OUT/prof.err"
wc -l "$OUT/prof.err.full" >> "$OUT/prof.err"; rm -f "$OUT/prof.err.full"
grep -h "naive" "$OUT"/prof/*/*kernel_stats.csv > "$OUT/prof_naive.csv" 2>/dev/null
rm -rf "$OUT/prof"
ls -la /tmp/glr_miopen_db_* > "$OUT/tmp_ls.txt" 2>&1
env | grep -i "miopen\|tmpdir\|home" > "$OUT/env.txt"
echo "[probe] done"
