#!/bin/bash
O=gpurun_out/r4e19; mkdir -p $O
for cap in 768 1536 3072; do
  SM3_BN_GRID_CAP=$cap timeout -k 10 200 python scratch/prof_detail.py 256 > $O/per_shape_cap$cap.txt 2>&1
  echo "== cap $cap"; grep "^bn_bwd_apply|\|^bn_act|\|sum of" $O/per_shape_cap$cap.txt | head -24
done
