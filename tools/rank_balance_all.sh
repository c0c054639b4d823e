#!/bin/bash
# usage: tools/rank_balance_all.sh <out.jsonl>  -- the tile deal's balance on the default scene (C1, C2) and the gorilla frame at 2, 4, 8 ranks, on one GPU
set -u
out=$1
: > $out
for shard in tiles samples; do
for n in 2 4 8; do
  timeout -k 10 300 python3 tools/rank_balance.py --scene final --width 480 --height 480 --spp 30 --depth 50 --mesh-depth 12 --world $n --reps 3 --shard $shard >> $out || exit 1
  timeout -k 10 300 python3 tools/rank_balance.py --scene final --width 480 --height 480 --spp 16 --depth 50 --mesh-depth 12 --world $n --reps 3 --shard $shard >> $out || exit 1
  timeout -k 10 300 python3 tools/rank_balance.py --scene mesh_gorilla --width 1920 --height 1080 --spp 30 --depth 50 --mesh-depth 12 --world $n --reps 3 --shard $shard >> $out || exit 1
done
done
python3 - $out <<'PY'
import json, sys
for l in open(sys.argv[1]):
    d = json.loads(l)
    print(d["scene"], d["frame"], d["shard"], "ranks", d["world"], "max", d["max_ms"], "mean", d["mean_ms"], "max/mean", d["imbalance_max_over_mean"], "rays max/mean", round(max(r["rays"] for r in d["ranks"]) / (sum(r["rays"] for r in d["ranks"]) / d["world"]), 4))
PY
