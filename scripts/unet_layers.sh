#!/bin/bash
# Per-layer time of the U-Net: kernel trace of scripts/time_encoder.py, the 17 conv dispatches of the last launch_unet.
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ul && rocprofv3 --kernel-trace --output-format csv -d /tmp/ul -- python $GRAFT_REPO_ROOT/scripts/time_encoder.py ${1:-2468} > /tmp/ul.log 2>&1
tail -1 /tmp/ul.log
python - <<'P'
import csv, glob
f = glob.glob('/tmp/ul/**/*kernel_trace.csv', recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if 'conv_kernel' in r['Kernel_Name'] or 'wino_kernel' in r['Kernel_Name']]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
last = rows[-17:]
names = "d0a d0 d1a d1 d2a d2 d3a d3 up0 u0a u0 up1 u1a u1 up2 u2a u2+fin".split()
# flop per image: HW^2 * taps * Cin * Cout * 2
spec = [(64,9,32,32),(64,9,32,32),(32,9,32,64),(32,9,64,64),(16,9,64,128),(16,9,128,128),(8,9,128,256),(8,9,256,256),
        (8,4,256,128),(16,9,256,128),(16,9,128,128),(16,4,128,64),(32,9,128,64),(32,9,64,64),(32,4,64,32),(64,9,64,32),(64,10,32,32)]
import sys
n_img = 3 * int(open('/tmp/ul.log').read().split('clouds ')[1].split(':')[0])
tot = 0
for nm, r, (hw, taps, ci, co) in zip(names, last, spec):
    us = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    fl = hw * hw * taps * ci * co * 2 * n_img
    tot += us
    print("%-4s %-44s grid %-14s %8.1f us  %6.1f TFLOP/s" % (nm, r['Kernel_Name'][10:50], r.get('Grid_Size_X','')+','+r.get('Grid_Size_Y','')+','+r.get('Grid_Size_Z',''), us, fl / us / 1e6))
print("total %.1f us" % tot)
P
