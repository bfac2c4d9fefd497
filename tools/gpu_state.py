#!/usr/bin/env python
"""What state is this GPU box in?  (bench.py's ``gpu_state`` object)

Boxes of the pool run the same binary 25-30 % apart on the latency- and VALU-bound kernels while the
HBM-bound ones hardly move.  This collects what can tell the classes apart:

* the driver's view: performance level, clock tables with the active level, power cap and draw,
  compute / memory partition mode (sysfs first, ``rocm-smi --json`` as a second source);
* the clock kernels really see: ``mgx_probe_clock`` (tools/probe/libmgx_probe.so) (s_memtime against the constant 100 MHz counter)
  on a nearly idle chip (one workgroup), on a chip whose every SIMD issues VALU, and straight after a
  pause (how fast the clock comes back).

    python tools/gpu_state.py            # prints one JSON object
"""
import ctypes
import glob
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _read(path):
    try:
        with open(path) as fh:
            return fh.read().strip()
    except OSError:
        return None


def _active_level(table):
    """'0: 132Mhz\n1: 2400Mhz *' -> {'levels': [...], 'active': '2400Mhz'}"""
    if not table:
        return None
    levels, active = [], None
    for line in table.splitlines():
        parts = line.replace("*", " *").split()
        if len(parts) >= 2:
            levels.append(parts[1])
            if "*" in parts:
                active = parts[1]
    return {"levels": levels, "active": active}


def sysfs_state():
    out = {}
    for dev in sorted(glob.glob("/sys/class/drm/card*/device")):
        if not os.path.exists(os.path.join(dev, "pp_dpm_sclk")):
            continue
        d = {
            "perf_level": _read(os.path.join(dev, "power_dpm_force_performance_level")),
            "sclk": _active_level(_read(os.path.join(dev, "pp_dpm_sclk"))),
            "mclk": _active_level(_read(os.path.join(dev, "pp_dpm_mclk"))),
            "fclk": _active_level(_read(os.path.join(dev, "pp_dpm_fclk"))),
            "socclk": _active_level(_read(os.path.join(dev, "pp_dpm_socclk"))),
            "compute_partition": _read(os.path.join(dev, "current_compute_partition")),
            "memory_partition": _read(os.path.join(dev, "current_memory_partition")),
            "gpu_busy_percent": _read(os.path.join(dev, "gpu_busy_percent")),
            "vbios": _read(os.path.join(dev, "vbios_version")),
        }
        for hw in glob.glob(os.path.join(dev, "hwmon", "hwmon*")):
            for key, name in (("power_cap_W", "power1_cap"), ("power_cap_max_W", "power1_cap_max"),
                              ("power_average_W", "power1_average"), ("power_input_W", "power1_input")):
                v = _read(os.path.join(hw, name))
                if v and v.isdigit():
                    d[key] = int(v) / 1e6
            v = _read(os.path.join(hw, "temp1_input"))
            if v and v.lstrip("-").isdigit():
                d["temp_C"] = int(v) / 1e3
        out[os.path.basename(os.path.dirname(dev))] = d
    return out


def smi_state(timeout=20):
    """rocm-smi's JSON for card 0 (trimmed to the fields that matter); {} when the tool fails."""
    try:
        raw = subprocess.run(
            ["rocm-smi", "--showclocks", "--showpower", "--showmaxpower", "--showperflevel", "--showcomputepartition",
             "--showmemorypartition", "--showtemp", "--showuse", "--json"],
            capture_output=True, text=True, timeout=timeout).stdout
        data = json.loads(raw[raw.index("{"):])
    except Exception as exc:                                     # noqa: BLE001 - a diagnostic, never fatal
        return {"error": repr(exc)[:200]}
    card = data.get("card0") or next(iter(data.values()), {})
    keep = {}
    for k, v in card.items():
        lk = k.lower()
        if any(t in lk for t in ("sclk", "mclk", "fclk", "socclk", "power", "performance", "partition", "temperature (sensor junction",
                                  "gpu use")):
            keep[k] = v
    return keep


def _probe_module():
    import importlib

    here = os.path.dirname(os.path.abspath(__file__))
    if here not in sys.path:
        sys.path.insert(0, here)
    return importlib.import_module("mgx_probe")


def clock_probe(dev=None):
    """Shader MHz by mgx_probe_clock (tools/probe): idle chip, loaded chip, and the first kernel after a pause."""
    index = 0 if dev is None else dev.index
    mgx_probe = _probe_module()

    def probe(wgs, iters):
        out = mgx_probe.clock(index, wgs, iters)
        return {"mhz": round(out[2], 1), "ms": round(out[3], 4)}

    res = {}
    probe(4096, 20000)                                           # wake the chip
    res["loaded"] = probe(8192, 60000)                           # every SIMD issuing VALU for ~tens of ms
    res["idle_one_workgroup"] = probe(1, 200000)
    res["short_loaded"] = probe(2048, 2000)                      # a kernel of tens of microseconds
    time.sleep(0.5)
    res["after_500ms_pause_short"] = probe(2048, 2000)
    res["then_loaded"] = probe(8192, 60000)
    return res


def own_pci_bus(timeout=20):
    """PCI address of the one GPU this process can open (rocm-smi lists only that one as card0, sysfs lists
    every GPU of the node)."""
    try:
        raw = subprocess.run(["rocm-smi", "--showbus", "--json"], capture_output=True, text=True, timeout=timeout).stdout
        data = json.loads(raw[raw.index("{"):])
        card = data.get("card0") or next(iter(data.values()), {})
        for k, v in card.items():
            if "pci" in k.lower():
                return str(v).strip().lower()
    except Exception:                                            # noqa: BLE001
        pass
    return None


def _card_of(bus):
    for dev in glob.glob("/sys/class/drm/card*/device"):
        if bus and os.path.basename(os.path.realpath(dev)).lower() == bus:
            return os.path.basename(os.path.dirname(dev))
    return None


def compact_state(dev=None):
    """The few numbers that tell one box of the pool from another, for bench.py's JSON line."""
    import threading

    out = {}
    sysfs = sysfs_state()
    bus = own_pci_bus()
    card = _card_of(bus) or ("card0" if "card0" in sysfs else next(iter(sysfs), None))
    mine = sysfs.get(card, {})
    out["card"] = card
    out["pci_bus"] = bus
    for k in ("perf_level", "compute_partition", "memory_partition", "power_cap_W", "vbios"):
        out[k] = mine.get(k)
    out["sclk_levels"] = (mine.get("sclk") or {}).get("levels")
    out["mclk_levels"] = (mine.get("mclk") or {}).get("levels")
    # other GPUs of the node that are clocked up right now (other tenants' jobs share the chassis)
    busy = 0
    for name, d in sysfs.items():
        act = ((d.get("sclk") or {}).get("active") or "0Mhz").lower().replace("mhz", "")
        if name != card and act.isdigit() and int(act) >= 1000:
            busy += 1
    out["gpus_on_node"] = len(sysfs)
    out["other_gpus_clocked_up"] = busy
    try:
        index = 0 if dev is None else dev.index
        mgx_probe = _probe_module()

        def probe(wgs, iters):
            res = mgx_probe.clock(index, wgs, iters)
            return round(res[2], 1), round(res[3], 3)

        probe(4096, 20000)
        seen = {}

        def watch():                                             # power and clock as the driver reports them, mid-probe
            time.sleep(0.15)
            now = sysfs_state().get(card, {})
            seen["power_W"] = now.get("power_input_W", now.get("power_average_W"))
            seen["sclk"] = (now.get("sclk") or {}).get("active")
            seen["temp_C"] = now.get("temp_C")

        th = threading.Thread(target=watch)
        th.start()
        out["shader_mhz_all_simds_busy"], out["probe_ms"] = probe(8192, 2000000)     # ~0.3 s of dependent FMAs on every SIMD
        th.join()
        out["under_probe"] = seen
        out["shader_mhz_one_workgroup"], _ = probe(1, 200000)
        out["shader_mhz_short_kernel"], out["short_kernel_ms"] = probe(2048, 2000)
        mem = mgx_probe.memory(index)
        out["memory_probe"] = {"ns_per_dependent_load": {"1GiB_hbm": round(mem[0], 1), "2MiB_l2": round(mem[1], 1),
                                                         "8KiB_first_level": round(mem[2], 1)},
                               "stream_read_GBs": round(mem[3], 1), "us_per_empty_launch": round(mem[4], 2),
                               "ns_per_instruction_112KiB_code": {"cold": round(mem[5], 2), "again": round(mem[6], 2)},
                               "ns_per_dependent_lds_read": round(mem[7], 1), "ns_per_barrier_256": round(mem[8], 1),
                               "ns_per_returning_atomic": round(mem[9], 1),
                               "ns_per_instruction_looping_over_code_of": {"16KiB": round(mem[10], 2), "32KiB": round(mem[11], 2),
                                                                            "48KiB": round(mem[12], 2), "64KiB": round(mem[13], 2)}}
    except Exception as exc:                                     # noqa: BLE001
        out["probe_error"] = repr(exc)[:200]
    return out


def gpu_state(dev=None, with_smi=True):
    state = {"sysfs": sysfs_state()}
    if with_smi:
        state["rocm_smi"] = smi_state()
    try:
        state["clock_probe"] = clock_probe(dev)
        state["sysfs_after_probe"] = {k: {"sclk": v.get("sclk", {}) and v["sclk"].get("active"),
                                          "mclk": v.get("mclk", {}) and v["mclk"].get("active"),
                                          "power_average_W": v.get("power_average_W")}
                                      for k, v in sysfs_state().items()}
    except Exception as exc:                                     # noqa: BLE001
        state["clock_probe"] = {"error": repr(exc)[:300]}
    return state


if __name__ == "__main__":
    if "--compact" in sys.argv:
        print(json.dumps(compact_state(), indent=1))
    else:
        print(json.dumps(gpu_state(), indent=1))
