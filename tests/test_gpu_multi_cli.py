"""The one-process-per-GPU front end (amplisolve_amd/multi.py) against the one-process executables: same tokens, same
files, byte for byte.  Rehearsed on ONE GPU: the ranks share cuda:0 and talk over gloo (RCCL needs a GPU per rank)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "amplisolve_amd", "bin")
G = "/root/repo/tests/golden"  # the goldens were taken with this directory literal (it decides the visit order, EE:794-841)


def free_port() -> int:
    """a TCP port nobody on this host listens on right now"""
    import socket

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def torchrun(world, port, args, env=None):
    port = free_port()  # the callers' literals only tell the launches apart when reading a failure
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), "-m", "amplisolve_amd.multi"] + args
    e = dict(os.environ, AMPLISOLVE_DIST_BACKEND="gloo", PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""), **(env or {}))
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT, env=e)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    return r


def strip_dates(text):
    return "\n".join(l for l in text.splitlines() if not l.startswith("##fileDate="))


@pytest.mark.parametrize("world", [2, 3, 4])
def test_error_estimation_sharded_equals_one_process(tmp_path, world):
    """5 normal files over 2, 3 and 4 processes.  The GPU box allows 6 processes on its card: pytest's own context, the launcher
    (torch.distributed.run opens the device too) and four ranks -- round 6 tried five and the pool's process guard ended the run.  The
    native mode below (no launcher) goes to five; the merge protocol itself runs with 8 ranks over gloo on the CPU (tests/test_dist_gloo.py)."""
    d = f"{G}/toy_subset"
    out = tmp_path / "multi"
    torchrun(world, 29541 + world, ["AmpliSolveErrorEstimation", f"panel_design={d}/panel.bed", "reference_genome=unused.fa", f"germline_dir={d}/NORMAL",
                                    "C_value=0.002", "coverage_cutoff=100", "default_error=0.01", f"output_dir={out}"],
             env={"AMPLISOLVE_REFBASES_FILE": f"{d}/refbases.txt"})
    assert (out / "positionSpecificNoise_0.0020.txt").read_text() == open(f"{d}/expected_positionSpecificNoise_0.0020.txt").read()
    names = os.listdir(out / "AmpliSolveErrorEstimation_interm_files")
    assert any(n.endswith("_germline_count_list_original.txt") for n in names) and any(n.endswith("_panelReferenceBases.txt") for n in names)


def test_error_estimation_more_processes_than_files(tmp_path):
    """2 normal files over 4 processes: two shards hold no file and contribute "no qualifying record" to the merge."""
    d = f"{G}/toy_subset"
    nd = tmp_path / "N2"
    nd.mkdir()
    for n in sorted(os.listdir(f"{d}/NORMAL"))[:2]:
        os.symlink(f"{d}/NORMAL/{n}", nd / n)
    args = [f"panel_design={d}/panel.bed", "reference_genome=unused.fa", f"germline_dir={nd}", "C_value=0.002", "coverage_cutoff=100",
            "default_error=0.01"]
    env = {"AMPLISOLVE_REFBASES_FILE": f"{d}/refbases.txt"}
    one, out = tmp_path / "one", tmp_path / "multi"
    r = subprocess.run([f"{BIN}/AmpliSolveErrorEstimation"] + args + [f"output_dir={one}"], capture_output=True, text=True,
                       env=dict(os.environ, AMPLISOLVE_STRICT_EXIT="1", **env))
    assert r.returncode == 0, r.stdout + r.stderr
    torchrun(4, 29549, ["AmpliSolveErrorEstimation"] + args + [f"output_dir={out}"], env=env)
    assert (out / "positionSpecificNoise_0.0020.txt").read_text() == (one / "positionSpecificNoise_0.0020.txt").read_text()


def test_error_estimation_sharded_edge_cases(tmp_path):
    """The synthetic edge-case panel (positions listed twice, absent records, NaN rates, quorum failures) at cov 1."""
    d = f"{G}/mini_edge"
    out = tmp_path / "multi"
    torchrun(3, 29551, ["AmpliSolveErrorEstimation", f"panel_design={d}/panel.bed", "reference_genome=unused.fa", f"germline_dir={d}/NORMAL",
                        "C_value=0.0005", "coverage_cutoff=1", "default_error=0.01", f"output_dir={out}"],
             env={"AMPLISOLVE_REFBASES_FILE": f"{d}/refbases.txt"})
    assert (out / "positionSpecificNoise_0.0005.txt").read_text() == open(f"{d}/expected_positionSpecificNoise_0.0005_cov1.txt").read()


@pytest.mark.parametrize("world", [2, 3, 4])
def test_variant_calling_sharded_equals_one_process(tmp_path, world):
    """3 tumour files (one without any call) over 2, 3 and 4 processes: Summary and every VCF as the one-process run."""
    d = f"{G}/toy_subset"
    table = f"{d}/expected_positionSpecificNoise_0.0020.txt"
    one = tmp_path / "one"
    r = subprocess.run([f"{BIN}/AmpliSolveVariantCalling", f"errorFile={table}", f"tumour_dir={d}/TUMOUR", f"output_dir={one}",
                        "coverage_cutoff=100", "p_value=0.05"], capture_output=True, text=True, env=dict(os.environ, AMPLISOLVE_STRICT_EXIT="1"))
    assert r.returncode == 0, r.stdout + r.stderr
    out = tmp_path / "multi"
    torchrun(world, 29561 + world, ["AmpliSolveVariantCalling", f"errorFile={table}", f"tumour_dir={d}/TUMOUR", f"output_dir={out}",
                                    "coverage_cutoff=100", "p_value=0.05"])
    assert (out / "Summary_Variant_Info.txt").read_text() == (one / "Summary_Variant_Info.txt").read_text()
    vcfs = sorted(n for n in os.listdir(one) if n.endswith(".vcf"))
    assert vcfs == sorted(n for n in os.listdir(out) if n.endswith(".vcf")) and len(vcfs) == 3
    for n in vcfs:
        assert strip_dates((out / n).read_text()) == strip_dates((one / n).read_text())
    assert not [n for n in os.listdir(out) if ".part" in n]
    assert os.path.exists(out / "AmpliSolveVariantCalling_interm_files" / "dummyVCF_1.vcf")


def test_variant_calling_middle_shard_without_calls(tmp_path):
    """A shard in the MIDDLE of the visit order whose only tumour file holds no call: its Summary part is empty, and the
    parts of the shards behind it must still arrive in Summary_Variant_Info.txt (an empty streambuf insert sets failbit)."""
    import ctypes as C
    import shutil

    from amplisolve_amd import host_lib

    d = f"{G}/toy_subset"
    table = f"{d}/expected_positionSpecificNoise_0.0020.txt"
    td = tmp_path / "TUM"
    td.mkdir()
    names = ["KA", "KB", "KC"]
    for n in names:  # placeholders first: the visit order is a function of the listed path strings only
        (td / f"{n}.PILEUP.ASEQ").write_text("x\n")
    buf = C.create_string_buffer(1 << 12)
    assert host_lib().ampli_host_sample_order(str(td).encode(), buf, len(buf)) == 3
    order = buf.value.decode().split()
    header = open(f"{d}/TUMOUR/T1.PILEUP.ASEQ").readline()
    shutil.copy(f"{d}/TUMOUR/T1.PILEUP.ASEQ", td / f"{order[0]}.PILEUP.ASEQ")
    (td / f"{order[1]}.PILEUP.ASEQ").write_text(header)  # header only: no line, no call
    shutil.copy(f"{d}/TUMOUR/T2.PILEUP.ASEQ", td / f"{order[2]}.PILEUP.ASEQ")
    one = tmp_path / "one"
    r = subprocess.run([f"{BIN}/AmpliSolveVariantCalling", f"errorFile={table}", f"tumour_dir={td}", f"output_dir={one}",
                        "coverage_cutoff=100", "p_value=0.05"], capture_output=True, text=True, env=dict(os.environ, AMPLISOLVE_STRICT_EXIT="1"))
    assert r.returncode == 0, r.stdout + r.stderr
    ref = (one / "Summary_Variant_Info.txt").read_text()
    samples = [l.split("\t")[0] for l in ref.splitlines()[1:]]
    assert order[0] in samples and order[2] in samples and order[1] not in samples
    out = tmp_path / "multi"
    torchrun(3, 29569, ["AmpliSolveVariantCalling", f"errorFile={table}", f"tumour_dir={td}", f"output_dir={out}",
                        "coverage_cutoff=100", "p_value=0.05"])
    assert (out / "Summary_Variant_Info.txt").read_text() == ref


def test_failing_shard_ends_the_job(tmp_path):
    """A shard that cannot read its input fails the whole launch promptly (no hang in a collective) with the reason."""
    d = f"{G}/toy_subset"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), "-m", "amplisolve_amd.multi", "AmpliSolveVariantCalling", f"errorFile={tmp_path}/missing_table.txt",
           f"tumour_dir={d}/TUMOUR", f"output_dir={tmp_path}/o", "coverage_cutoff=100", "p_value=0.05"]
    e = dict(os.environ, AMPLISOLVE_DIST_BACKEND="gloo", PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300, cwd=ROOT, env=e)
    assert r.returncode != 0
    assert "failed" in r.stderr and "Something went wrong" in (r.stdout + r.stderr)


def test_without_a_launcher_it_is_the_one_process_pipeline(tmp_path):
    d = f"{G}/toy_subset"
    out = tmp_path / "solo"
    e = dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""), AMPLISOLVE_REFBASES_FILE=f"{d}/refbases.txt")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        e.pop(k, None)
    r = subprocess.run([sys.executable, "-m", "amplisolve_amd.multi", "AmpliSolveErrorEstimation", f"panel_design={d}/panel.bed",
                        "reference_genome=unused.fa", f"germline_dir={d}/NORMAL", "C_value=0.002", "coverage_cutoff=100", "default_error=0.01",
                        f"output_dir={out}"], capture_output=True, text=True, timeout=300, cwd=ROOT, env=e)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert (out / "positionSpecificNoise_0.0020.txt").read_text() == open(f"{d}/expected_positionSpecificNoise_0.0020.txt").read()


# ---- the executables' own multi-GPU mode: no Python, RCCL transport of libamplisolve_hip.so (ampli_comm_*) ----
def test_native_rccl_transport_both_command_lines_on_a_communicator_of_one(tmp_path):
    """AMPLISOLVE_FORCE_NATIVE_DIST=1 sends a one-process run through the sharded pipeline over a REAL RCCL communicator
    of size 1 (all a one-GPU box offers): id-file rendezvous, reduce-scatter, grouped send/recv all-to-all, all-gather,
    all-reduce and the row-count gather all execute; outputs byte-identical to the plain run / the reference goldens."""
    d = f"{G}/toy_subset"
    env = dict(os.environ, AMPLISOLVE_STRICT_EXIT="1", AMPLISOLVE_FORCE_NATIVE_DIST="1", AMPLISOLVE_REFBASES_FILE=f"{d}/refbases.txt")
    out = tmp_path / "native"
    r = subprocess.run([f"{BIN}/AmpliSolveErrorEstimation", f"panel_design={d}/panel.bed", "reference_genome=unused.fa", f"germline_dir={d}/NORMAL",
                        "C_value=0.002", "coverage_cutoff=100", "default_error=0.01", f"output_dir={out}"], capture_output=True, text=True,
                       timeout=300, env=env)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "shard 1/1" in r.stdout
    assert (out / "positionSpecificNoise_0.0020.txt").read_text() == open(f"{d}/expected_positionSpecificNoise_0.0020.txt").read()
    assert not os.path.exists(out / ".amplisolve_rccl_id")  # rank 0 removes the rendezvous file once everybody has read it

    table = f"{d}/expected_positionSpecificNoise_0.0020.txt"
    one, nat = tmp_path / "one", tmp_path / "native_vc"
    for o, e in ((one, dict(os.environ, AMPLISOLVE_STRICT_EXIT="1")), (nat, env)):
        r = subprocess.run([f"{BIN}/AmpliSolveVariantCalling", f"errorFile={table}", f"tumour_dir={d}/TUMOUR", f"output_dir={o}",
                            "coverage_cutoff=100", "p_value=0.05"], capture_output=True, text=True, timeout=300, env=e)
        assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert (nat / "Summary_Variant_Info.txt").read_text() == (one / "Summary_Variant_Info.txt").read_text()
    assert not [n for n in os.listdir(nat) if ".part" in n]


def test_native_shard_without_rank_zero_fails_loudly(tmp_path):
    """Rank 1 of 2 with nobody publishing the communicator id: a bounded wait, a message naming the file, exit status 1."""
    d = f"{G}/toy_subset"
    env = dict(os.environ, AMPLISOLVE_WORLD_SIZE="2", AMPLISOLVE_RANK="1", AMPLISOLVE_RCCL_TIMEOUT="2", AMPLISOLVE_DEVICE="0",
               AMPLISOLVE_REFBASES_FILE=f"{d}/refbases.txt")
    r = subprocess.run([f"{BIN}/AmpliSolveErrorEstimation", f"panel_design={d}/panel.bed", "reference_genome=unused.fa", f"germline_dir={d}/NORMAL",
                        "C_value=0.002", "coverage_cutoff=100", "default_error=0.01", f"output_dir={tmp_path}/o"], capture_output=True, text=True,
                       timeout=120, env=env)
    assert r.returncode == 1
    assert "timed out waiting for rank 0's id file" in r.stdout


# ---- the native mode with SEVERAL ranks on one GPU: RCCL refuses that, so a shared-memory test double of the few RCCL entry points
# the HIP library binds (tests/fake_rccl/fake_rccl.cpp, AMPLISOLVE_RCCL_LIB) stands in for the wire.  What this checks is everything
# around the wire: NativeShard's buffers, counts and chunk order, the id-file rendezvous between real processes, shard 0 as the
# writer, the row-count prefix of the Summary, empty shards.
@pytest.fixture(scope="module")
def fake_rccl(tmp_path_factory):
    so = tmp_path_factory.mktemp("fake_rccl") / "libfake_rccl.so"
    r = subprocess.run(["hipcc", "-O1", "-fPIC", "-shared", "-std=c++17", "-o", str(so), os.path.join(ROOT, "tests", "fake_rccl", "fake_rccl.cpp"), "-lrt"],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    return str(so)


def native_run(world, exe, args, out, fake, env=None):
    os.makedirs(out, exist_ok=True)
    procs = []
    for k in range(world):
        e = dict(os.environ, AMPLISOLVE_WORLD_SIZE=str(world), AMPLISOLVE_RANK=str(k), AMPLISOLVE_DEVICE="0", AMPLISOLVE_RCCL_LIB=fake,
                 AMPLISOLVE_ID_FILE=os.path.join(str(out), "rendezvous.id"), AMPLISOLVE_RCCL_TIMEOUT="120", **(env or {}))
        procs.append(subprocess.Popen([f"{BIN}/{exe}"] + args + [f"output_dir={out}"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, env=e))
    outs = [p.communicate(timeout=600)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n=====\n".join(o[-2000:] for o in outs)
    return outs


@pytest.mark.parametrize("world", [2, 3, 4, 5])
def test_native_mode_several_ranks_error_estimation(tmp_path, fake_rccl, world):
    d = f"{G}/toy_subset"
    out = tmp_path / "native"
    outs = native_run(world, "AmpliSolveErrorEstimation", [f"panel_design={d}/panel.bed", "reference_genome=unused.fa", f"germline_dir={d}/NORMAL", "C_value=0.002",
                                                           "coverage_cutoff=100", "default_error=0.01"], out, fake_rccl, env={"AMPLISOLVE_REFBASES_FILE": f"{d}/refbases.txt"})
    assert (out / "positionSpecificNoise_0.0020.txt").read_text() == open(f"{d}/expected_positionSpecificNoise_0.0020.txt").read()
    assert all(f"shard {k + 1}/{world}" in outs[k] for k in range(world))
    assert not os.path.exists(out / "rendezvous.id")


def test_native_mode_edge_cases_and_empty_shards(tmp_path, fake_rccl):
    """the synthetic edge-case panel at cov 1 over 3 ranks, and 2 normal files over 4 ranks (two shards hold no file)"""
    d = f"{G}/mini_edge"
    out = tmp_path / "edge"
    native_run(3, "AmpliSolveErrorEstimation", [f"panel_design={d}/panel.bed", "reference_genome=unused.fa", f"germline_dir={d}/NORMAL", "C_value=0.0005",
                                                "coverage_cutoff=1", "default_error=0.01"], out, fake_rccl, env={"AMPLISOLVE_REFBASES_FILE": f"{d}/refbases.txt"})
    assert (out / "positionSpecificNoise_0.0005.txt").read_text() == open(f"{d}/expected_positionSpecificNoise_0.0005_cov1.txt").read()
    d = f"{G}/toy_subset"
    nd = tmp_path / "N2"
    nd.mkdir()
    for n in sorted(os.listdir(f"{d}/NORMAL"))[:2]:
        os.symlink(f"{d}/NORMAL/{n}", nd / n)
    args = [f"panel_design={d}/panel.bed", "reference_genome=unused.fa", f"germline_dir={nd}", "C_value=0.002", "coverage_cutoff=100", "default_error=0.01"]
    env = {"AMPLISOLVE_REFBASES_FILE": f"{d}/refbases.txt"}
    one = tmp_path / "one"
    r = subprocess.run([f"{BIN}/AmpliSolveErrorEstimation"] + args + [f"output_dir={one}"], capture_output=True, text=True, env=dict(os.environ, AMPLISOLVE_STRICT_EXIT="1", **env))
    assert r.returncode == 0, r.stdout + r.stderr
    native_run(4, "AmpliSolveErrorEstimation", args, tmp_path / "four", fake_rccl, env=env)
    assert (tmp_path / "four" / "positionSpecificNoise_0.0020.txt").read_text() == (one / "positionSpecificNoise_0.0020.txt").read_text()


@pytest.mark.parametrize("world", [2, 3, 5])
def test_native_mode_several_ranks_variant_calling(tmp_path, fake_rccl, world):
    d = f"{G}/toy_subset"
    table = f"{d}/expected_positionSpecificNoise_0.0020.txt"
    one = tmp_path / "one"
    r = subprocess.run([f"{BIN}/AmpliSolveVariantCalling", f"errorFile={table}", f"tumour_dir={d}/TUMOUR", f"output_dir={one}", "coverage_cutoff=100", "p_value=0.05"],
                       capture_output=True, text=True, env=dict(os.environ, AMPLISOLVE_STRICT_EXIT="1"))
    assert r.returncode == 0, r.stdout + r.stderr
    out = tmp_path / "native"
    os.makedirs(out)
    procs = []
    for k in range(world):
        e = dict(os.environ, AMPLISOLVE_WORLD_SIZE=str(world), AMPLISOLVE_RANK=str(k), AMPLISOLVE_DEVICE="0", AMPLISOLVE_RCCL_LIB=fake_rccl,
                 AMPLISOLVE_ID_FILE=str(out / "rendezvous.id"))
        procs.append(subprocess.Popen([f"{BIN}/AmpliSolveVariantCalling", f"errorFile={table}", f"tumour_dir={d}/TUMOUR", f"output_dir={out}", "coverage_cutoff=100",
                                       "p_value=0.05"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, env=e))
    outs = [p.communicate(timeout=600)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n=====\n".join(o[-2000:] for o in outs)
    assert (out / "Summary_Variant_Info.txt").read_text() == (one / "Summary_Variant_Info.txt").read_text()
    vcfs = sorted(n for n in os.listdir(one) if n.endswith(".vcf"))
    assert vcfs == sorted(n for n in os.listdir(out) if n.endswith(".vcf"))
    for n in vcfs:
        assert strip_dates((out / n).read_text()) == strip_dates((one / n).read_text())
    assert not [n for n in os.listdir(out) if ".part" in n]


def _write_id_file(path, world, nonce=0, age_s=0, magic=b"AMPLRCC2"):
    """an id file as ampli_comm_create writes it (csrc/ampli_comm.hip, struct IdFile) with a dead communicator id"""
    import struct
    import time

    with open(path, "wb") as f:
        f.write(magic + struct.pack("<iiQq", world, 0, nonce, int(time.time()) - age_s) + b"/fake_rccl_dead_run".ljust(128, b"\0"))


def test_stale_id_file_is_ignored_and_the_job_runs(tmp_path, fake_rccl):
    """A run that died left its id file behind (old time stamp, dead id).  Round 2 let ranks > 0 read it on their first poll and
    wait in ncclCommInitRank for ever; now rank 0 replaces it and the readers do not accept what is older than the job."""
    d = f"{G}/toy_subset"
    out = tmp_path / "native"
    os.makedirs(out)
    _write_id_file(out / "rendezvous.id", world=2, age_s=3600)
    native_run(2, "AmpliSolveErrorEstimation", [f"panel_design={d}/panel.bed", "reference_genome=unused.fa", f"germline_dir={d}/NORMAL", "C_value=0.002",
                                                "coverage_cutoff=100", "default_error=0.01"], out, fake_rccl, env={"AMPLISOLVE_REFBASES_FILE": f"{d}/refbases.txt"})
    assert (out / "positionSpecificNoise_0.0020.txt").read_text() == open(f"{d}/expected_positionSpecificNoise_0.0020.txt").read()


@pytest.mark.parametrize("case", ["fresh_dead_id", "other_world", "other_nonce", "old_format"])
def test_id_file_of_another_job_fails_loudly_not_for_ever(tmp_path, fake_rccl, case):
    """Rank 1 of 2 alone with an id file that is not its job's.  A file of another world size, another launch (nonce) or
    another format is rejected by name; one that looks right but whose communicator is dead gets as far as
    ncclCommInitRank, which is bounded by the same timeout.  Either way: exit status 1 within seconds, a message saying why."""
    import time

    d = f"{G}/toy_subset"
    out = tmp_path / "o"
    os.makedirs(out)
    idf = out / "rendezvous.id"
    env = dict(os.environ, AMPLISOLVE_WORLD_SIZE="2", AMPLISOLVE_RANK="1", AMPLISOLVE_RCCL_TIMEOUT="3", AMPLISOLVE_DEVICE="0",
               AMPLISOLVE_RCCL_LIB=fake_rccl, AMPLISOLVE_ID_FILE=str(idf), AMPLISOLVE_REFBASES_FILE=f"{d}/refbases.txt", AMPLISOLVE_JOB_NONCE="77")
    if case == "fresh_dead_id":
        _write_id_file(idf, world=2, nonce=77)
        expect = "ncclCommInitRank did not complete within 3 s"
    elif case == "other_world":
        _write_id_file(idf, world=4, nonce=77)
        expect = "written for 4 ranks, this job has 2"
    elif case == "other_nonce":
        _write_id_file(idf, world=2, nonce=78)
        expect = "belongs to another launch"
    else:
        open(idf, "wb").write(b"/fake_rccl_dead_run".ljust(128, b"\0"))  # the bare 128-byte id of round 2
        expect = "not an id file of this library version"
    t0 = time.time()
    r = subprocess.run([f"{BIN}/AmpliSolveErrorEstimation", f"panel_design={d}/panel.bed", "reference_genome=unused.fa", f"germline_dir={d}/NORMAL",
                        "C_value=0.002", "coverage_cutoff=100", "default_error=0.01", f"output_dir={out}"], capture_output=True, text=True,
                       timeout=120, env=env)
    try:
        assert r.returncode == 1, r.stdout[-2000:] + r.stderr[-2000:]
        assert expect in r.stdout, r.stdout[-2000:]
        assert time.time() - t0 < 60
    finally:
        if os.path.exists("/dev/shm/fake_rccl_dead_run"):  # the test double's segment for the dead id
            os.unlink("/dev/shm/fake_rccl_dead_run")


def test_native_launcher_ends_the_job_when_one_shard_fails(tmp_path, fake_rccl):
    """tools/launch_native.sh: shard 1 cannot read one of its files and exits 1 before its first collective; the other ranks are
    already inside theirs (RCCL -- and the test double -- would wait for ever).  The launcher must stop them and return 1
    promptly instead of waiting for every pid in turn."""
    import shutil
    import time

    d = f"{G}/toy_subset"
    nd = tmp_path / "N"
    shutil.copytree(f"{d}/NORMAL", nd)
    names = sorted(os.listdir(nd))
    os.remove(nd / names[-1])
    os.symlink("/nonexistent/target.PILEUP.ASEQ", nd / names[-1])  # listed by the directory scan, unreadable for the shard that owns it
    env = dict(os.environ, AMPLISOLVE_RCCL_LIB=fake_rccl, AMPLISOLVE_DEVICE="0", AMPLISOLVE_RCCL_TIMEOUT="60", AMPLISOLVE_REFBASES_FILE=f"{d}/refbases.txt",
               TMPDIR=str(tmp_path))
    t0 = time.time()
    r = subprocess.run([os.path.join(ROOT, "tools", "launch_native.sh"), "3", "AmpliSolveErrorEstimation", f"panel_design={d}/panel.bed", "reference_genome=unused.fa",
                        f"germline_dir={nd}", "C_value=0.002", "coverage_cutoff=100", "default_error=0.01", f"output_dir={tmp_path}/o"],
                       capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 1, r.stdout[-3000:] + r.stderr[-3000:]
    assert "stopping the others" in r.stderr
    assert time.time() - t0 < 120
    assert not [n for n in os.listdir(tmp_path) if n.startswith("amplisolve_rccl_id.")]


def test_native_mode_cohort_outside_the_exactness_envelope(tmp_path, fake_rccl):
    """Three ranks on a cohort whose threshold sums depend on the order of addition (coverage cut-off 1, depths of tens of millions):
    the merged sums raise the envelope flag on every rank, the WRITER alone streams the whole cohort again in the reference's own order
     while the others wait at the barrier, and the table is the reference's, byte for byte."""
    from oracle import pyoracle as orc
    from tests.helpers import write_envelope_panel

    if not os.path.exists(orc.REF_EE_DRIVER):
        pytest.skip("oracle/_ref/ee_ref_driver is absent")
    d = tmp_path / "panel"
    d.mkdir()
    write_envelope_panel(d, 5, S=11)
    (d / "o").mkdir()
    r = subprocess.run([orc.REF_EE_DRIVER, "p.bed", "r.txt", "d.txt", "N", "0.002", "1", "o"], capture_output=True, text=True, cwd=d)
    assert r.returncode == 0, r.stderr[-400:]
    name = [n for n in os.listdir(d / "o") if n.startswith("positionSpecificNoise_")][0]
    out = tmp_path / "native"
    outs = native_run(3, "AmpliSolveErrorEstimation", [f"panel_design={d}/p.bed", "reference_genome=unused.fa", f"germline_dir={d}/N", "C_value=0.002",
                                                      "coverage_cutoff=1", "default_error=0.01"], out, fake_rccl,
                      env={"AMPLISOLVE_REFBASES_FILE": f"{d}/r.txt", "AMPLISOLVE_LIST_DIR_AS": "N", "AMPLISOLVE_CHUNK_BYTES": "40000"})
    assert all("summing again in the reference's order" in o for o in outs)
    assert (out / name).read_bytes() == (d / "o" / name).read_bytes()
