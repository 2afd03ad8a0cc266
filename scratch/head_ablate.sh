#!/bin/bash
# timing-only ablations of head_reg_kernel (results wrong by construction for ABL != 0)
for rot in 3 -1; do for abl in 0 1 2 3; do
  echo "== DCLR_HR_ROT=$rot DCLR_HR_ABL=$abl"
  DCLR_HR_ROT=$rot DCLR_HR_ABL=$abl timeout -k 10 200 python scratch/head_probe.py 2>&1 | grep -E "^rows|reg " | head -4
done; done
