#!/bin/bash
# Flow-embedding timing ablations (scratch/libdeepclr_fabl.so = ab_build.sh fabl flow16.hip -DDCLR_ABLATION; results WRONG):
# DCLR_FLOW_ABL bits: 1 no gather round trips, 2 weight fragments from one address (L1 hits), 4 no phase A, 8 no weight loads.
cd "$(dirname "$0")/.."
for a in 0 1 2 8 9 4 12; do echo -n "DCLR_FLOW_ABL=$a: "; DCLR_FLOW_ABL=$a DCLR_LIB=scratch/libdeepclr_fabl.so python scratch/flow_probe.py 2>&1 | grep "80 pairs"; done
