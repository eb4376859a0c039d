"""Generate tests/golden/hip_net_pins.json ON an MI355X (see tests/net_pins.py):
    python tests/golden/make_hip_net_pins.py [out.json]        (default: gpurun_out/hip_net_pins.json -- copy it to tests/golden/)"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE)); sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import net_pins  # noqa: E402

out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.dirname(HERE)), 'gpurun_out', 'hip_net_pins.json')
pins = net_pins.compute()
os.makedirs(os.path.dirname(out), exist_ok=True)
with open(out, 'w') as fh:
    json.dump({'device': 'gfx950 (MI355X)', 'what': 'crc32 of NNetWrapper.process policy / value float32 rows; third entry: row 0 evaluated alone has the same bits',
               'pins': pins}, fh, indent=1, sort_keys=True)
print(json.dumps(pins))
