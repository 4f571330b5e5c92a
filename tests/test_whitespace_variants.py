"""Whitespace the reference's reader does not care about, and ours must not either.  The reference reads a data line with
`fscanf("%1000[^\\n]\\n")` + `sscanf("%s\\t%s\\t...\\t%d")` (EE:1113-1149, VC:721-752): any run of blanks / tabs separates two columns, a
`\\t` in the format matches none or many, leading blanks and blank lines vanish in the previous line's trailing `\\n` directive, `%d`
takes a `+`, a carriage return in front of the newline ends the last integer, columns beyond the fifteenth are never looked at, and
the last line needs no newline.  Two CPU checks tie our reader to that by transitivity: (1) the reference's own code
(oracle/_ref/ee_ref_driver, compiled where the sources lie) writes the same table, byte for byte, for a directory of files roughened
that way as for the clean files; (2) our reader hands over the same records, extras, RD side list and line statistics for both --
through the one-pass tokeniser and through the plain one.  (Ours == the reference on the clean files: tests/test_gpu_cli.py,
tools/fuzz_cli_vs_reference.py.)"""
import os
import subprocess

import numpy as np
import pytest

from amplisolve_amd.hostio import HostCohort
from oracle import pyoracle as orc
from tests.helpers import write_fresh_panel


def _roughen(text, rng, final_newline):
    head, *lines = text.split("\n")
    if lines and lines[-1] == "":
        lines.pop()
    out = [head + "\n"]
    for ln in lines:
        tok = ln.split("\t")
        m = rng.random()
        sep, eol, lead = "\t", "\n", ""
        if m < 0.25:
            pass
        elif m < 0.35:
            sep = " "
        elif m < 0.45:
            sep = "\t\t"
        elif m < 0.52:
            sep = " \t "
        elif m < 0.60:
            eol = "\r\n"
        elif m < 0.68:
            lead = "  \t"
        elif m < 0.76:
            i = int(rng.integers(6, 15))
            tok[i] = "+" + tok[i]
        elif m < 0.84:
            tok += ["extra", "7", "columns"]
        elif m < 0.92:
            tok[-1] += " \t "
        else:
            tok[2], tok[3] = "rs12345", "0.25"  # a dbSNP id and a frequency where the toy files carry dots
        out.append(lead + sep.join(tok) + eol)
        b = rng.random()
        if b < 0.04:
            out.append("\n")
        elif b < 0.07:
            out.append(" \t\r\n")
    s = "".join(out)
    return s if final_newline else s.rstrip("\r\n \t")


@pytest.mark.parametrize("seed", [11, 12])
def test_whitespace_the_reference_ignores_changes_nothing(tmp_path, monkeypatch, seed):
    rng = np.random.default_rng(seed)
    clean, rough = tmp_path / "clean", tmp_path / "rough"
    clean.mkdir(), rough.mkdir()
    write_fresh_panel(clean, seed, S=6, amplicons=4)
    write_fresh_panel(rough, seed, S=6, amplicons=4)
    for k, f in enumerate(sorted(os.listdir(rough / "N"))):
        p = rough / "N" / f
        p.write_bytes(_roughen(p.read_text(), rng, final_newline=k % 3 != 1).encode())
    assert (rough / "N" / sorted(os.listdir(rough / "N"))[0]).read_bytes() != (clean / "N" / sorted(os.listdir(clean / "N"))[0]).read_bytes()
    # (1) the reference's own reader
    if os.path.exists(orc.REF_EE_DRIVER):
        tables = []
        for d in (clean, rough):
            (d / "o").mkdir()
            r = subprocess.run([orc.REF_EE_DRIVER, "p.bed", "r.txt", "d.txt", "N", "0.002", "100", "o"], capture_output=True, text=True, cwd=d)
            assert r.returncode == 0, r.stdout[-400:] + r.stderr[-400:]
            name = [n for n in os.listdir(d / "o") if n.startswith("positionSpecificNoise_")]
            assert len(name) == 1
            tables.append((d / "o" / name[0]).read_bytes())
        assert tables[0] == tables[1] and len(tables[0]) > 10_000
    # (2) ours, both tokenisers
    seen = []
    for parser in ("plain", None):
        monkeypatch.setenv("AMPLISOLVE_PARSER", parser) if parser else monkeypatch.delenv("AMPLISOLVE_PARSER")
        for d in (clean, rough):
            monkeypatch.chdir(d)  # the same directory literal both times: the visit order hangs on it (a1)
            co = HostCohort("p.bed", "N", refbases_file="r.txt", keep_line_no=True)
            st = co.stats()
            assert st["malformed"] == 0 and st["lines"] > 1000
            seen.append((st, co.recs.tobytes(), co.line_no.tobytes(), co.E, list(co.names)))
            co.close()
    assert all(s == seen[0] for s in seen[1:])
