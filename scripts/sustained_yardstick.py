"""Yardstick only (not a product path): SUSTAINED rates — launches back to back, no host synchronisation in between, seconds per
shape — of PyTorch-ROCm's library GEMM (hipBLASLt behind torch.nn.functional.linear, bias only) and of libssp2vit's persistent
kernel (tools/gemm_bench with GEMM_SUSTAIN) on the step's projection shapes, with the card's power and shader clock sampled from
sysfs beside both.  Round 2's yardstick waited for every launch (so does gemm_bench's median): the card then idles between two
launches and the power controller sees bursts, which the prune step — 1 300 launches back to back — never offers.

  python3 scripts/sustained_yardstick.py [seconds per run, default 2.5]  ->  JSON lines on stdout"""
import glob, json, os, subprocess, sys, threading, time
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "2ssp-x-vit_amd", "csrc", "tools", "gemm_bench.bin")
SECONDS = float(sys.argv[1]) if len(sys.argv) > 1 else 2.5


def _sysfs():
    """power (uW) and sclk files of the card this process computes on: sysfs shows every card of the host (eight here, other
    tenants' included), so the card is picked by the PCI address HIP reports for device 0; all of them only if that fails"""
    power, sclk = [], []
    cards = sorted(glob.glob("/sys/class/drm/card*/device"))
    try:
        pr = torch.cuda.get_device_properties(0)
        addr = f"{pr.pci_domain_id:04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}.0"
        mine = [c for c in cards if os.path.realpath(c).endswith(addr)]
        cards = mine or cards
    except Exception:
        pass
    for card in cards:
        for name in ("power1_average", "power1_input"):
            power += glob.glob(os.path.join(card, "hwmon", "hwmon*", name))
        sclk += glob.glob(os.path.join(card, "hwmon", "hwmon*", "freq1_input"))
    return power, sclk


class Sampler(threading.Thread):
    def __init__(self, period=0.02):
        super().__init__(daemon=True)
        self.power_files, self.sclk_files = _sysfs()
        self.period, self.rows, self.stop_flag = period, [], False

    @staticmethod
    def _read(files):
        best = None
        for f in files:
            try:
                v = float(open(f).read().strip())
                best = v if best is None else max(best, v)     # several cards visible in sysfs: the busy one
            except (OSError, ValueError):
                pass
        return best

    def run(self):
        while not self.stop_flag:
            self.rows.append((time.time(), self._read(self.power_files), self._read(self.sclk_files)))
            time.sleep(self.period)

    def window(self, t0, t1):
        rows = [r for r in self.rows if t0 + 0.3 * (t1 - t0) <= r[0] <= t1]      # the settled part of the run
        pw = [r[1] for r in rows if r[1] is not None]
        ck = [r[2] for r in rows if r[2] is not None]
        return {"power_w": round(sum(pw) / len(pw) / 1e6, 1) if pw else None, "power_w_max": round(max(pw) / 1e6, 1) if pw else None,
                "sclk_mhz": round(sum(ck) / len(ck) / 1e6, 1) if ck else None, "samples": len(rows)}


def library(M, N, K, sampler):
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(1)
    a = (torch.rand(M, K, device=dev, generator=g) * 2 - 1).to(torch.bfloat16)
    w = ((torch.rand(N, K, device=dev, generator=g) * 2 - 1) * 0.05).to(torch.bfloat16)
    b = torch.zeros(N, device=dev, dtype=torch.bfloat16)
    fn = lambda: torch.nn.functional.linear(a, w, b)
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); [fn() for _ in range(20)]; e1.record(); e1.synchronize()
    n = max(50, int(SECONDS / (e0.elapsed_time(e1) / 20 * 1e-3)))
    per = 50
    chunks = (n + per - 1) // per
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(chunks + 1)]
    t0 = time.time()
    ev[0].record()
    for c in range(chunks):
        for _ in range(per): fn()
        ev[c + 1].record()
    torch.cuda.synchronize()
    t1 = time.time()
    cm = [ev[c].elapsed_time(ev[c + 1]) / per for c in range(chunks)]
    us = sum(cm) / len(cm) * 1e3
    # the per-launch waited form of round 2's yardstick, for the same tensors
    ts = []
    for _ in range(20):
        a0, a1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a0.record(); fn(); a1.record(); a1.synchronize(); ts.append(a0.elapsed_time(a1))
    ts.sort()
    return {"sustained_us": round(us, 1), "sustained_tflops": round(2.0 * M * N * K / (us * 1e-6) / 1e12, 1), "launches": chunks * per,
            "first50_us": round(cm[0] * 1e3, 1), "last50_us": round(cm[-1] * 1e3, 1),
            "waited_median_us": round(ts[10] * 1e3, 1), "waited_min_us": round(ts[0] * 1e3, 1), **sampler.window(t0, t1)}


def ours(M, N, K, epi, sampler, est_us):
    n = max(50, int(SECONDS / (est_us * 1e-6)))
    env = dict(os.environ, GEMM_SUSTAIN=str(n))
    t0 = time.time()
    out = subprocess.run([BENCH, str(M), str(N), str(K), str(epi), "20"], env=env, capture_output=True, text=True, timeout=300).stdout
    t1 = time.time()
    res = {"raw": [l.strip() for l in out.splitlines() if "median" in l or "sustained" in l]}
    for l in out.splitlines():
        if "sustained" in l:
            res["sustained_us"] = float(l.split("):")[1].split("us")[0])
            res["sustained_tflops"] = float(l.split("->")[1].split("TFLOP")[0])
            # the sustained loop is the LAST part of the process's life: sample the last SECONDS of it
            res.update(sampler.window(max(t0, t1 - SECONDS * 1.1), t1))
    return res


def main():
    sampler = Sampler()
    sampler.start()
    print(json.dumps({"sysfs_power": sampler.power_files, "sysfs_sclk": sampler.sclk_files, "device": torch.cuda.get_device_name(0)}), flush=True)
    #        name        M      N     K    (our epilogue, label) ...
    shapes = [("QKV", 63040, 2304, 768, [(10, "bias, bf16 out")]),
              ("out-proj", 63040, 768, 768, [(10, "bias, bf16 out"), (11, "bias + fp32 residual")]),
              ("fc1", 63040, 3072, 768, [(10, "bias, bf16 out"), (12, "bias + erf-GELU")]),
              ("fc2", 63040, 768, 3072, [(10, "bias, bf16 out"), (11, "bias + fp32 residual")]),
              ("fc1 layer-major", 315200, 3072, 768, [(12, "bias + erf-GELU")]),
              ("H/14 QKV", 82240, 3840, 1280, [(10, "bias, bf16 out")]),
              ("H/14 fc1", 82240, 5120, 1280, [(10, "bias, bf16 out"), (12, "bias + erf-GELU")])]
    for name, M, N, K, epis in shapes:
        lib = library(M, N, K, sampler)
        row = {"shape": name, "M": M, "N": N, "K": K, "library(linear+bias)": lib}
        torch.cuda.empty_cache()
        for epi, label in epis:
            row[f"ours epi {epi} ({label})"] = ours(M, N, K, epi, sampler, lib["sustained_us"] * 1.3)
        print(json.dumps(row), flush=True)
    sampler.stop_flag = True


if __name__ == "__main__":
    main()
