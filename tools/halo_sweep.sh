#!/bin/bash
# How short can the halo in front of a rank's block be before the chain states at the cuts stop being reproduced by the warm-up?
# Eight gloo ranks on ONE device (the driver's `bench.py --gpus 8` job at 240 and 690 scans), halo = 12 ... 2 steps.
# usage (GPU box): bash tools/halo_sweep.sh > gpurun_out/r06_halo_sweep.txt
for scans in 240 690; do
  for halo in 12 10 8 6 5 4 3 2; do
    out=$(timeout 300 python bench.py --gpus 8 --same-device --backend gloo --scans $scans --split-halo $halo --steps 1 --warmup 1 --no-cpu --no-extras 2>/dev/null | grep '^{' | tail -1)
    python - "$scans" "$halo" <<PY
import json, sys
try:
    d = json.loads('''$out''')
    sp = d["config"]["split"]
    print(f"scans {sys.argv[1]:>4s} halo {sys.argv[2]:>2s} steps: loaded by rank 0 {sp['scans_loaded_by_rank0']}, own {sp['own']}, chains walked again at a cut (all ranks, all steps of the run) {sp['chains_rewalked_at_boundary_all_ranks']}, ms/step {d['ms_per_step']:.2f}")
except Exception as e:
    print(f"scans {sys.argv[1]} halo {sys.argv[2]}: no line ({e})")
PY
  done
done
