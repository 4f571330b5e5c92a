"""TEST INFRASTRUCTURE -- CPU restatement of the pileup step upstream of AmpliSolve (BAM -> .PILEUP.ASEQ), used only by tests/
to check amplisolve_amd's computeCounts.  Never imported by the product.

What it restates: the PILEUP mode of ASEQ (Romanel et al., BMC Med Genomics 2015), which the reference ships as the source-less
binary Pre-compiled_binaries/computeCounts and describes in /root/reference/Execution_examples.md:16-46
(`vcf= bam= threads= mbq= mrq= mdc= out=`).  PARITY UNPINNED: that binary is a Mach-O executable that cannot run here and the
reference holds no BAM fixture, only outputs of the step (Toy_data/*.PILEUP.ASEQ), whose observable properties this restatement
reproduces: the header line, tab-separated columns `chr pos dbsnp MAF ref alt A C G T RD Ars Crs Grs Trs`, RD = A+C+G+T on every
line, one line per LISTED position in list order (a position listed twice is written twice with identical counts), positions
below the depth cut-off left out.  Semantics (htslib's pileup defaults, which ASEQ drives):
  * reads: skipped when unmapped / secondary / QC-fail / duplicate (0x4|0x100|0x200|0x400) or MAPQ < mrq;
  * columns: only bases aligned by M, = or X; deletions and reference skips give nothing to the positions they span;
  * bases: A/C/G/T only (N and IUPAC codes are not counted), base quality >= mbq;
  * X = bases on both strands, Xrs = bases of reads with flag 0x10; a line is written when RD >= mdc.
Pure Python: small inputs only.
"""
import gzip
import struct

SEQ_CODE = "=ACMGRSVTWYHKDBN"
CIGAR_OPS = "MIDNSHP=X"


def read_bam(path):
    """(reference names, records); a record = dict(ref_id, pos (0-based), mapq, flag, cigar [(op char, length)], seq, qual)."""
    raw = gzip.open(path, "rb").read()  # BGZF = concatenated gzip members
    assert raw[:4] == b"BAM\1"
    l_text, = struct.unpack_from("<i", raw, 4)
    o = 8 + l_text
    n_ref, = struct.unpack_from("<i", raw, o)
    o += 4
    refs = []
    for _ in range(n_ref):
        l_name, = struct.unpack_from("<i", raw, o)
        refs.append(raw[o + 4:o + 4 + l_name - 1].decode())
        o += 4 + l_name + 4
    recs = []
    while o + 4 <= len(raw):
        bs, = struct.unpack_from("<i", raw, o)
        r = raw[o + 4:o + 4 + bs]
        o += 4 + bs
        if len(r) < bs or bs < 32:
            break
        ref_id, pos, l_name, mapq, _bin, n_cigar, flag, l_seq = struct.unpack_from("<iiBBHHHi", r, 0)
        c = 32 + l_name
        cigar = []
        for i in range(n_cigar):
            v, = struct.unpack_from("<I", r, c + 4 * i)
            cigar.append((CIGAR_OPS[v & 15] if (v & 15) < 9 else "?", v >> 4))
        s = c + 4 * n_cigar
        seq = "".join(SEQ_CODE[(r[s + (i >> 1)] >> (4 if i % 2 == 0 else 0)) & 15] for i in range(l_seq))
        qual = list(r[s + (l_seq + 1) // 2: s + (l_seq + 1) // 2 + l_seq])
        recs.append(dict(ref_id=ref_id, pos=pos, mapq=mapq, flag=flag, cigar=cigar, seq=seq, qual=qual))
    return refs, recs


def pileup(refs, recs, positions, mbq, mrq):
    """positions: iterable of (chrom, 1-based pos).  Returns {(chrom, pos): [A, C, G, T, Ars, Crs, Grs, Trs]}."""
    want = {}
    for c, p in positions:
        want.setdefault((c, p), [0] * 8)
    for r in recs:
        if r["ref_id"] < 0 or r["pos"] < 0 or (r["flag"] & (0x4 | 0x100 | 0x200 | 0x400)) or r["mapq"] < mrq:
            continue
        if sum(n for op, n in r["cigar"] if op in "MIS=X") != len(r["seq"]):
            continue  # malformed: the host scanner drops it too
        chrom = refs[r["ref_id"]]
        rev = bool(r["flag"] & 0x10)
        refpos, q = r["pos"] + 1, 0
        for op, n in r["cigar"]:
            if op in "M=X":
                for j in range(n):
                    cell = want.get((chrom, refpos + j))
                    b = "ACGT".find(r["seq"][q + j])
                    if cell is not None and b >= 0 and r["qual"][q + j] >= mbq:
                        cell[b] += 1
                        if rev:
                            cell[4 + b] += 1
                refpos += n
                q += n
            elif op in "IS":
                q += n
            elif op in "DN":
                refpos += n
    return want


def aseq_text(lines, counts, mdc):
    """lines: [(chrom, pos, id, ref, alt)] in list order."""
    out = ["chr\tpos\tdbsnp\tMAF\tref\talt\tA\tC\tG\tT\tRD\tArs\tCrs\tGrs\tTrs"]
    for c, p, i, ref, alt in lines:
        v = counts.get((c, p), [0] * 8)
        rd = sum(v[:4])
        if rd >= mdc:
            out.append("\t".join([c, str(p), i, ".", ref, alt] + [str(x) for x in v[:4]] + [str(rd)] + [str(x) for x in v[4:]]))
    return "\n".join(out) + "\n"
