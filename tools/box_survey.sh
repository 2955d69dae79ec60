#!/bin/bash
# One survey record per GPU box: tools/box_info.sh, tools/l2_persist (do clean lines survive a kernel boundary here?), tools/box_probe.py (does the next-weight
# prefetch pay here?).  Usage (GPU box): bash tools/box_survey.sh TAG  -> gpurun_out/box/TAG.{info,l2,probe}
TAG=${1:-x}
mkdir -p gpurun_out/box
bash tools/box_info.sh > gpurun_out/box/$TAG.info 2>&1
hipcc -O2 --offload-arch=gfx950 tools/l2_persist.hip -o /tmp/l2_persist && timeout -k 10 120 /tmp/l2_persist > gpurun_out/box/$TAG.l2 2>&1
timeout -k 10 300 python tools/box_probe.py > gpurun_out/box/$TAG.probe 2> gpurun_out/box/$TAG.probe.err
python - <<PY
import json
d=json.loads(open("gpurun_out/box/$TAG.probe").read().strip().splitlines()[-1])
print("$TAG step", d["step"], "fc2 clock", d["fc2_M720_clock_ghz"], "copy", d["copy_tb_s"])
PY
grep -E "same XCD|another XCD|untouched" gpurun_out/box/$TAG.l2 | head -3
