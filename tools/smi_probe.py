#!/usr/bin/env python3
"""What can a process read about the GPU's clock and power WITHOUT starting another program (runs ON THE GPU BOX)?
Lists the amdgpu hwmon / sysfs files and tries librocm_smi64 / libamd_smi through ctypes; prints a few samples idle and
(with --load) beside a running bench.  The answer decides how bench.py samples sclk / power (yolov3/gpu_telemetry.py)."""
import ctypes
import glob
import os
import sys
import time


def rd(p):
    try:
        with open(p) as fh:
            return fh.read().strip()
    except OSError as e:
        return "<%s>" % e.__class__.__name__


def main():
    for card in sorted(glob.glob("/sys/class/drm/card*/device")):
        if not os.path.exists(os.path.join(card, "vendor")):
            continue
        print("==", card, rd(os.path.join(card, "vendor")), rd(os.path.join(card, "device")))
        for f in ("pp_dpm_sclk", "pp_dpm_mclk", "gpu_busy_percent", "current_link_speed"):
            print("  ", f, "=", rd(os.path.join(card, f)).replace("\n", " | "))
        for hw in sorted(glob.glob(os.path.join(card, "hwmon", "hwmon*"))):
            for f in sorted(os.listdir(hw)):
                p = os.path.join(hw, f)
                if os.path.isfile(p) and (f.startswith(("power", "freq", "temp1")) or f == "name"):
                    print("  ", os.path.relpath(p, card), "=", rd(p))
    for name in ("librocm_smi64.so", "/opt/rocm/lib/librocm_smi64.so", "libamd_smi.so", "/opt/rocm/lib/libamd_smi.so"):
        try:
            lib = ctypes.CDLL(name)
            print("loaded", name)
        except OSError as e:
            print("cannot load", name, e)
            continue
        if "rocm_smi" in name:
            rc = lib.rsmi_init(ctypes.c_uint64(0))
            print("  rsmi_init", rc)
            n = ctypes.c_uint32(0)
            print("  rsmi_num_monitor_devices", lib.rsmi_num_monitor_devices(ctypes.byref(n)), n.value)

            class Freqs(ctypes.Structure):
                _fields_ = [("has_deep_sleep", ctypes.c_bool), ("num_supported", ctypes.c_uint32), ("current", ctypes.c_uint32),
                            ("frequency", ctypes.c_uint64 * 33)]
            for _ in range(3):
                f = Freqs()
                rc = lib.rsmi_dev_gpu_clk_freq_get(ctypes.c_uint32(0), ctypes.c_int(0), ctypes.byref(f))
                pw = ctypes.c_uint64(0)
                ptype = ctypes.c_int(0)
                rc2 = lib.rsmi_dev_power_get(ctypes.c_uint32(0), ctypes.byref(pw), ctypes.byref(ptype)) if hasattr(lib, "rsmi_dev_power_get") else -1
                cur = f.frequency[f.current] if rc == 0 and f.current < 33 else None
                print("  sclk rc", rc, "n", f.num_supported, "cur", f.current, cur, " power rc", rc2, pw.value, ptype.value)
                time.sleep(0.2)
            break


if __name__ == "__main__":
    main()
    sys.stdout.flush()
