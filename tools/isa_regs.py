"""Registers / scratch / occupancy per kernel of a device-only assembly file (hipcc -S --offload-device-only):
python3 tools/isa_regs.py file.s [substring]"""
import re, subprocess, sys
want = sys.argv[2] if len(sys.argv) > 2 else ""
cur = None
info = {}
for line in open(sys.argv[1]):
    m = re.match(r"^(_Z\w+):", line)
    if m:
        cur = m.group(1)
        continue
    m = re.match(r"^; (NumVgprs|ScratchSize|Occupancy): (\d+)", line)
    if m and cur:
        info.setdefault(cur, {})[m.group(1)] = int(m.group(2))
names = list(info)
dem = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
for raw, name in zip(names, dem):
    name = name.replace("(anonymous namespace)::", "")
    name = re.sub(r"\(.*", "", name)
    if want in name:
        print(name, info[raw])
