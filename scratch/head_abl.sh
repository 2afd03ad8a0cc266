#!/bin/bash
# Head kernel timing probes: product build vs the H16_ABL builds of scratch/ab_build.sh (habl1: weights re-read from one
# fragment, habl2: no weight loads, habl4: no LDS activation reads, habl6: neither). Results are WRONG in the probes.
cd "$(dirname "$0")/.."
echo "product:"; python scratch/head_probe16.py
for a in 1 2 4 6; do echo "H16_ABL=$a:"; DCLR_LIB=scratch/libdeepclr_habl$a.so python scratch/head_probe16.py; done
