"""Round 6: what the HOST side of the one-worker-per-GPU scheduler costs per step at G = 1, 2, 4, 8 workers, measured on a
box with ONE GPU (geomodeling.devices = [0] * G: the workers share it).

  idle-GPU rows   a 4096^2 mosaic in 256 tiles of 256^2: the kernels take microseconds, a step is the host side alone --
                  parent wall time per request (messages to G workers sent and collected by one thread), the parent's own
                  CPU time, every worker's wall time between a request's arrival and its reply
  real rows       BASELINE configs[4] (65536^2 in 256 tiles of 4096^2): device-bound -- ms_per_step must stay what one worker
                  needs (the G workers share one GPU here), whatever G

Every row is `python bench.py --workload mosaic --gpus G ...` in a process of its own; the JSON lines go to stdout, a table to
stderr.  MEASUREMENTS.md section 12.1 is made from this."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def row(G, mosaic_side, side, steps, extra=()):
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "mosaic", "--gpus", str(G), "--mosaic-side",
           str(mosaic_side), "--side", str(side), "--steps", str(steps), "--warmup", "3", "--no-e2e"] + list(extra)
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT)
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    if res.returncode != 0 or not lines:
        sys.stderr.write("FAILED G=%d: %s\n" % (G, res.stderr[-800:]))
        return None
    print(lines[0], flush=True)
    return json.loads(lines[0])


def main():
    workers = [int(a) for a in sys.argv[1:]] or [1, 2, 4, 8]
    fmt = "%-22s %2s %10s %10s %12s %12s %12s %12s %8s\n"
    sys.stderr.write(fmt % ("rows", "G", "ms/step", "dev ms", "parent us", "parent cpu", "worker us max", "worker cpu", "cores"))
    for label, mosaic_side, side, steps, extra in (("idle GPU 256 x 256^2", 4096, 256, 200, ()),
                                                   ("idle, no pinning", 4096, 256, 200, ("--no-affinity",)),
                                                   ("real 256 x 4096^2", 65536, 4096, 20, ())):
        for G in workers:
            out = row(G, mosaic_side, side, steps, extra)
            if out is None:
                continue
            h = out["host"]
            sys.stderr.write(fmt % (label, G, out["ms_per_step"], out["device_ms_per_step"], h["parent_us_per_step"],
                                    h["parent_cpu_us_per_step"], max(h["worker_host_us_per_step"]),
                                    max(h["worker_host_cpu_us_per_step"]), h["worker_cpus"]))


if __name__ == "__main__":
    main()
