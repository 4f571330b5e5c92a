"""computeCounts (BAM -> .PILEUP.ASEQ) on a synthetic amplicon BAM: end to end through the executable, and the counting kernel alone
through the C ABI (HIP events; algorithmic bytes = the uncompressed alignment records, each read once, plus 8 bytes of record
offset per read).  usage: python tools/pileup_bench.py [reads] [read_len]"""
import ctypes as C
import os
import struct
import subprocess
import sys
import tempfile
import time
import zlib

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from amplisolve_amd import Context

N = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
L = int(sys.argv[2]) if len(sys.argv) > 2 else 120
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rng = np.random.default_rng(1)
n_amp = 400
amp_start = 10_000 + np.arange(n_amp) * 2_000  # one chromosome, amplicons 2 kb apart, every read of an amplicon starts at its start
refs = [("chr1", 10_000_000)]
text = b"@HD\tVN:1.6\tSO:coordinate\n@SQ\tSN:chr1\tLN:10000000\n"
header = b"BAM\1" + struct.pack("<i", len(text)) + text + struct.pack("<i", 1) + struct.pack("<i", 5) + b"chr1\0" + struct.pack("<i", 10_000_000)
name = b"r\0"
rec_len = 32 + len(name) + 4 + (L + 1) // 2 + L
dt = np.dtype([("bs", "<i4"), ("ref", "<i4"), ("pos", "<i4"), ("lname", "u1"), ("mapq", "u1"), ("bin", "<u2"), ("ncig", "<u2"), ("flag", "<u2"), ("lseq", "<i4"),
               ("nref", "<i4"), ("npos", "<i4"), ("tlen", "<i4"), ("name", "S2"), ("cig", "<u4"), ("seq", "u1", ((L + 1) // 2,)), ("qual", "u1", (L,))])
assert dt.itemsize == rec_len + 4
a = np.zeros(N, dt)
amp = np.sort(rng.integers(0, n_amp, N))
a["bs"], a["ref"], a["pos"], a["lname"], a["mapq"], a["bin"], a["ncig"] = rec_len, 0, amp_start[amp] - 1, len(name), 60, 4680, 1
a["flag"], a["lseq"], a["nref"], a["npos"], a["name"], a["cig"] = (rng.integers(0, 2, N) * 0x10).astype(np.uint16), L, -1, -1, name, (L << 4)
codes = np.array([1, 2, 4, 8], np.uint8)[rng.integers(0, 4, (N, 2 * ((L + 1) // 2)))]
a["seq"] = (codes[:, 0::2] << 4) | codes[:, 1::2]
a["qual"] = rng.integers(15, 41, (N, L)).astype(np.uint8)
raw = header + a.tobytes()
tmp = tempfile.mkdtemp(prefix="pileup_bench_")
bam = os.path.join(tmp, "S.bam")
t0 = time.time()
with open(bam, "wb") as f:
    for o in range(0, len(raw), 65280):
        chunk = raw[o:o + 65280]
        co = zlib.compressobj(1, zlib.DEFLATED, -15)
        data = co.compress(chunk) + co.flush()
        f.write(b"\x1f\x8b\x08\x04\0\0\0\0\0\xff\x06\0BC\x02\0" + struct.pack("<H", len(data) + 25) + data + struct.pack("<II", zlib.crc32(chunk) & 0xffffffff, len(chunk)))
    f.write(bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000"))
vcf = os.path.join(tmp, "v.txt")
with open(vcf, "w") as f:
    for s in amp_start:
        f.write("".join(f"chr1\t{p}\t.\t.\t.\t.\t.\t.\n" for p in range(s, s + L)))
print(f"synthetic BAM: {N} reads x {L} bp on {n_amp} amplicons, {len(raw) / 1e6:.0f} MB of records, {os.path.getsize(bam) / 1e6:.0f} MB on disk (written in {time.time() - t0:.1f} s)", flush=True)

for threads in (4, 16):
    t0 = time.time()
    r = subprocess.run([os.path.join(ROOT, "amplisolve_amd", "bin", "computeCounts"), f"vcf={vcf}", f"bam={bam}", f"threads={threads}", "mbq=20", "mrq=20", "mdc=20", f"out={tmp}"],
                       capture_output=True, text=True, env=dict(os.environ, AMPLISOLVE_TIMING="1"))
    wall = time.time() - t0
    assert r.returncode == 0, r.stdout + r.stderr
    print(f"computeCounts threads={threads}: wall {wall:.2f} s = {N / wall / 1e6:.2f} M reads/s = {len(raw) / wall / 1e9:.2f} GB/s of records;  {r.stderr.strip()}", flush=True)

# the kernel alone
ctx = Context(0)
d_bam = torch.frombuffer(bytearray(raw + bytes(64)), dtype=torch.uint8).cuda()  # the kernel copies whole 16-byte pieces
off = torch.from_numpy(len(header) + np.arange(N, dtype=np.uint64).astype(np.int64) * (rec_len + 4)).cuda()
keys = torch.from_numpy(np.concatenate([(np.int64(s) + np.arange(L, dtype=np.int64)) for s in amp_start])).cuda()  # ref id 0 << 32 | pos
counts = torch.zeros((keys.numel(), 8), dtype=torch.int32, device="cuda")
stats = torch.zeros(2, dtype=torch.int64, device="cuda")
call = lambda: ctx._check(ctx.lib.ampli_pileup_count(ctx.h, d_bam.data_ptr(), off.data_ptr(), N, keys.data_ptr(), keys.numel(), 20, 20, counts.data_ptr(), stats.data_ptr()))
for _ in range(2):
    call()
e0, e1 = ctx.event(), ctx.event()
ctx.record(e0)
for _ in range(10):
    call()
ctx.record(e1)
torch.cuda.synchronize()
ms = ctx.elapsed_ms(e0, e1) / 10
b = len(raw) + 8 * N
print(f"pileup_count_kernel: {ms * 1e3:.0f} us per launch = {b / ms / 1e6:.0f} GB/s of algorithmic bytes ({b / 1e6:.0f} MB), {N * L / ms / 1e6:.1f} G bases/s, "
      f"{int(stats[1].item()) // 12} counter updates per launch", flush=True)
ctx.close()
