#!/bin/bash
# what kind of box is this?  (the flaky-suite hunt: partition modes, clocks, firmware)
rocminfo 2>/dev/null | grep -i "Marketing Name\|Compute Unit\|Max Clock\|Uuid\|Node:" | grep -v "^$" | tr -s ' ' | head -24
rocm-smi --showcomputepartition --showmemorypartition --showvbios --showfwinfo 2>/dev/null | grep -v "^$\|=====" | head -40
python3 - <<'PY'
import torch
p = torch.cuda.get_device_properties(0)
print("torch:", p.name, "CUs", p.multi_processor_count, "mem GiB", round(p.total_memory / 2**30, 1), "devices", torch.cuda.device_count())
PY
