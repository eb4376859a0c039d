"""Clock and power of the GPU while (a) bare MFMAs run on zero / random fp16 operands and (b) the headline search launch runs -- run ON
the GPU box:  python tools/power_trace.py [--git HASH]   ->  gpurun_out/r04_power_trace.json

Why: tools/ubench/mfma_rate.hip sustains 2.40-2.47 PFLOP/s on zero operands but 1.96-2.04 (16x16x32) / 1.70 (32x32x16) on random
fp16 data, while MI355X_MICROARCH.md quotes 2.495 PFLOP/s measured.  If the difference is the power-management clock (DVFS) the
shader clock sampled DURING each run shows it: achieved TFLOP/s / (1 024 SIMDs x 1 024 FLOP/clk) = the clock the MFMAs ran at,
and the telemetry clock / power next to it says whether the chip was at its power limit.

Telemetry: amdgpu sysfs hwmon (freq1_input = sclk in Hz, power1_average / power1_input in uW) sampled every 50 ms, else
`rocm-smi --showclocks --showpower --json` (slower: ~0.3 s per sample), else `amd-smi metric`."""
import argparse
import glob
import json
import os
import re
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def find_hwmon():
    """hwmon directory of the GPU this process computes on (the host has several: match HIP device 0's PCI address; if that
    fails, the card whose gpu_busy_percent is highest while a short burst of work runs here)"""
    import torch
    try:
        pr = torch.cuda.get_device_properties(0)
        bdf = '%04x:%02x:%02x.0' % (pr.pci_domain_id, pr.pci_bus_id, pr.pci_device_id)
        hits = glob.glob('/sys/bus/pci/devices/%s/hwmon/hwmon*' % bdf)
        if hits:
            return hits[0]
    except Exception:                                            # noqa: BLE001
        pass
    x = torch.randn(8192, 8192, device='cuda', dtype=torch.float16)
    best, busy = None, -1
    for _ in range(20):
        x @ x
    for d in sorted(glob.glob('/sys/class/drm/card*/device')):
        b = read_int(os.path.join(d, 'gpu_busy_percent'))
        hw = glob.glob(os.path.join(d, 'hwmon', 'hwmon*'))
        if b is not None and hw and b > busy:
            best, busy = hw[0], b
    torch.cuda.synchronize()
    return best


def read_int(path):
    try:
        return int(open(path).read().strip())
    except (OSError, ValueError):
        return None


class Sampler(threading.Thread):
    def __init__(self):
        super().__init__(daemon=True)
        self.hw = find_hwmon()
        self.samples, self.stop_ev = [], threading.Event()
        self.kind = 'sysfs hwmon ' + self.hw if self.hw else None
        if self.hw is None:
            for tool in ('rocm-smi', 'amd-smi'):
                if subprocess.run(['which', tool], stdout=subprocess.PIPE).returncode == 0:
                    self.kind = tool
                    break

    def once(self):
        t = time.time()
        if self.hw:
            f = read_int(os.path.join(self.hw, 'freq1_input'))
            p = read_int(os.path.join(self.hw, 'power1_average'))
            if p is None:
                p = read_int(os.path.join(self.hw, 'power1_input'))
            return {'t': t, 'sclk_mhz': None if f is None else f / 1e6, 'power_w': None if p is None else p / 1e6}
        if self.kind == 'rocm-smi':
            r = subprocess.run(['rocm-smi', '-d', '0', '--showclocks', '--showpower', '--json'], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL)
            try:
                d = list(json.loads(r.stdout.decode()).values())[0]
            except Exception:                                   # noqa: BLE001
                return {'t': t}
            sclk = next((v for k, v in d.items() if 'sclk' in k.lower()), None)
            pw = next((v for k, v in d.items() if 'power' in k.lower() and 'w' in k.lower()), None)
            m = re.search(r'(\d+)\s*Mhz', str(sclk), re.I)
            return {'t': t, 'sclk_mhz': float(m.group(1)) if m else None, 'power_w': float(pw) if pw not in (None, 'N/A') else None, 'raw': None}
        if self.kind == 'amd-smi':
            r = subprocess.run(['amd-smi', 'metric', '-g', '0', '-c', '-p', '--json'], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL)
            txt = r.stdout.decode(errors='replace')
            mc = re.search(r'"gfx_0".*?"clk":\s*\{?\s*"value":\s*(\d+)', txt, re.S) or re.search(r'"clk":\s*"?(\d+)', txt)
            mp = re.search(r'"socket_power":\s*\{?\s*"value":\s*(\d+)', txt, re.S) or re.search(r'"socket_power":\s*"?(\d+)', txt)
            return {'t': t, 'sclk_mhz': float(mc.group(1)) if mc else None, 'power_w': float(mp.group(1)) if mp else None}
        return {'t': t}

    def run(self):
        while not self.stop_ev.is_set():
            self.samples.append(self.once())
            self.stop_ev.wait(0.05)


def summarise(samples, t0, t1):
    xs = [s for s in samples if t0 + 1.0 <= s['t'] <= t1 - 0.2]             # skip the ramp
    out = {'samples': len(xs)}
    for k in ('sclk_mhz', 'power_w'):
        v = [s[k] for s in xs if s.get(k) is not None]
        if v:
            out[k] = {'mean': round(sum(v) / len(v), 1), 'min': round(min(v), 1), 'max': round(max(v), 1)}
    return out


def search_loop(seconds):
    """the headline launch (connect4, 2048 games x 100 simulations, azg_search_f16) held for `seconds`; returns the algorithmic TFLOP/s"""
    import torch
    import bench
    from alphazero_general_amd.engine import DeviceEngine
    from alphazero_general_amd.envs.connect4 import Game
    from alphazero_general_amd.nnet import CONNECT4_NET_ARGS, NNetWrapper
    torch.manual_seed(0)
    net = NNetWrapper(Game, CONNECT4_NET_ARGS, device='cuda:0', dtype=torch.float16); net.refresh()
    B, sims = 2048, 100
    e = DeviceEngine(0, B, cpuct=4.0, fpu_reduction=0.4, add_root_noise=True, add_root_temp=True, seed=0, sims_hint=sims, example_capacity=B * 43 * 2 * 2)
    for _ in range(3):
        net._hip.search(e, sims); e.advance(True)
    torch.cuda.synchronize()
    t0 = time.time(); n = 0
    while time.time() - t0 < seconds:
        net._hip.search(e, sims); e.advance(True); n += 1
        if n % 8 == 0:
            e.clear_outputs()
        torch.cuda.synchronize()
    e.counters()                                                 # (raises on a sticky device error)
    dt = time.time() - t0
    return n, dt, bench.net_flops_per_leaf(Game, net.args) * B * sims * n / dt / 1e12


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--git', default='?')
    ap.add_argument('--seconds', type=float, default=5.0)
    a = ap.parse_args()
    sm = Sampler()
    cap = read_int(os.path.join(sm.hw, 'power1_cap')) if sm.hw else None
    out = {'git': a.git, 'telemetry': sm.kind, 'power_cap_w': None if cap is None else cap / 1e6, 'note': 'clock_from_rate_mhz = TFLOP/s / (1 024 SIMDs x 1 024 FLOP per clock): the clock the MFMA pipes '
           'must have run at if they issued back to back; sclk / power = device telemetry sampled during the run', 'cases': []}
    sm.start()
    hold = os.path.join(ROOT, 'tools', 'ubench', 'mfma_hold')
    time.sleep(1.5)
    t_idle0, t_idle1 = time.time() - 1.5, time.time() + 0.2
    for shape, zero in ((16, 1), (16, 0), (32, 1), (32, 0)):
        t0 = time.time()
        r = subprocess.run([hold, str(shape), str(zero), str(a.seconds)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)
        t1 = time.time()
        line = [l for l in r.stdout.decode().splitlines() if l.startswith('{')]
        rec = json.loads(line[-1]) if line else {'error': r.stderr.decode(errors='replace')[-300:]}
        if 'tflops' in rec:
            rec['clock_from_rate_mhz'] = round(rec['tflops'] * 1e12 / (1024 * 1024) / 1e6, 1)
            rec['frac_of_2500'] = round(rec['tflops'] / 2500.0, 4)
        rec.update(summarise(sm.samples, t0, t1))
        out['cases'].append(rec)
        print(rec, flush=True)
        time.sleep(1.0)
    t0 = time.time() + 0.0
    n, dt, tf = search_loop(a.seconds)
    t1 = time.time()
    rec = {'shape': 'k_tower2<...,SearchArgs<C4>> (azg_search_f16, connect4 2048 x 100)', 'operands': 'network activations / weights', 'launches': n,
           'ms_per_launch': round(dt * 1e3 / n, 3), 'algorithmic_tflops': round(tf, 1)}
    rec.update(summarise(sm.samples, t1 - dt, t1))
    out['cases'].append(rec)
    print(rec, flush=True)
    sm.stop_ev.set(); sm.join(2)
    out['idle'] = summarise(sm.samples, t_idle0 - 1.0, t_idle1 + 0.2)
    os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
    with open(os.path.join(ROOT, 'gpurun_out', 'r04_power_trace.json'), 'w') as fh:
        json.dump(out, fh, indent=1)


if __name__ == '__main__':
    main()
