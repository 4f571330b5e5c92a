"""ON THE GPU BOX: what makes the kernel's teardown of a config-4-sized command line take 0.09-0.17 s after main() (r05_exit_probe.log)?
A child process does ONE kind of host-side work, with or without a context of the HIP library open (nothing is ever launched or
copied), and leaves with _exit; printed: the time from the child's last line to the parent's wait4 returning.
kinds: none | touch (1.5 GB of fresh anonymous memory written) | threads (400 threads started and joined) | maps (the 1024 files
mapped, read and unmapped one by one) | parse1 / parse16 (the host library's ingest, 1 / 16 threads) | parse16_then_ctx (the context
opened AFTER the parsing) | sleepT (idle for T seconds with the context open: `... 64 age`)."""
import mmap, os, sys, time, subprocess, tempfile, threading
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
P = 100000


def open_ctx():
    from amplisolve_amd import Context
    return Context(0)


if len(sys.argv) > 1 and sys.argv[1] == "child":
    d, kind, with_ctx = sys.argv[2], sys.argv[3], sys.argv[4] == "1"
    import numpy as np
    from amplisolve_amd._lib import host_lib
    from amplisolve_amd.hostio import HostCohort, CHUNK_FN
    ctx = open_ctx() if with_ctx and kind != "parse16_then_ctx" else None
    t0 = time.time()
    if kind == "touch":
        a = np.empty(1500 << 20, np.uint8); a[::4096] = 1
    elif kind == "threads":
        for _ in range(25):
            th = [threading.Thread(target=lambda: None) for _ in range(16)]
            [t.start() for t in th]; [t.join() for t in th]
    elif kind == "maps":
        tot = 0
        for f in sorted(os.listdir(d + "/N")):
            with open(os.path.join(d, "N", f), "rb") as fh:
                m = mmap.mmap(fh.fileno(), 0, prot=mmap.PROT_READ); tot += m[::4096].count(b"c"); m.close()
    elif kind.startswith("sleep"):
        time.sleep(float(kind[5:]))
    elif kind.startswith("parse"):
        H = host_lib()
        co = HostCohort(d + "/panel.bed", refbases_file=d + "/refbases.txt")
        H.ampli_host_stream_chunks(co.h, (d + "/N").encode(), 1 if kind == "parse1" else 16, 0, 128 << 20, CHUNK_FN(lambda *a: 0), None)
        if kind == "parse16_then_ctx":
            ctx = open_ctx()
    t1 = time.time()
    sys.stderr.write(f"EPOCH {t1:.6f} work {t1 - t0:.3f}\n")
    sys.stderr.flush()
    os._exit(0)

S = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
from amplisolve_amd._lib import host_lib
H = host_lib()
d = tempfile.mkdtemp(prefix="ampli_exit_cpu_")
H.ampli_host_synth_write_panel((d + "/panel.bed").encode(), (d + "/refbases.txt").encode(), P, 0xA3F15017)
H.ampli_host_synth_write_aseq((d + "/N").encode(), b"N", P, S, 0, 0xA3F15017, 2000, 0, 0)
CASES = (("none", 0), ("none", 1), ("touch", 0), ("touch", 1), ("threads", 1), ("maps", 1), ("parse1", 1), ("parse16", 0), ("parse16", 1), ("parse16_then_ctx", 1))
if len(sys.argv) > 2 and sys.argv[2] == "age":  # only: how old is the context when the process leaves?
    CASES = tuple((f"sleep{t}", 1) for t in (0, 0.02, 0.05, 0.1, 0.15, 0.2, 0.3, 0.5, 1.0, 0.05, 0, 0.3))
for rep in range(2):
    for kind, with_ctx in CASES:
        p = subprocess.Popen([sys.executable, __file__, "child", d, kind, str(with_ctx)], stderr=subprocess.PIPE, stdout=subprocess.DEVNULL, text=True)
        _, status, ru = os.wait4(p.pid, 0)
        t_end = time.time()
        err = p.stderr.read()
        ep = float(err.split("EPOCH ")[1].split()[0])
        print(f"{kind:17s} GPU context {'yes' if with_ctx else 'no '}: after the child's last line {t_end - ep:.4f} s   ({err.strip().splitlines()[-1].split(' ', 2)[2]}; maxrss {ru.ru_maxrss / 1024:.0f} MB)", flush=True)
subprocess.run(["rm", "-rf", d])
