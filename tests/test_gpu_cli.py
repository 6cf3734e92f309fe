"""GPU test of the drop-in surface: `lash sketch` (C++ CLI over the C ABI) writes {o}_sketches.bin / {o}_files.json /
{o}_parameters.json exactly as the reference does (utils.rs:567-580, main.rs:254-276); the decompressed .bin must be
the concatenation of the oracle's images in list order (BASELINE.json configs[0] plumbing on the bundled fixtures)."""
import json
import os
import subprocess

import numpy as np
import pytest

import host_lib as H
import oracle_lib as O
from fastx import read_fastx

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
ALGO = {"hmh": O.HMH, "hll": O.HLL, "ull": O.ULL}


@pytest.mark.parametrize("algo,k,p,extra", [("hmh", 16, 10, []), ("hll", 21, 14, []), ("ull", 16, 12, ["--batch-mb", "1"]),
                                            ("hmh", 11, 10, ["-t", "3", "--batch-mb", "1"]),
                                            # the multi-GPU path: three workers (here all on GPU 0) finish batches out of
                                            # order, the writer must still emit images in file order
                                            ("ull", 19, 11, ["--devices", "0,0,0", "--batch-mb", "1"])])
def test_lash_sketch_cli_outputs(tmp_path, algo, k, p, extra):
    names = ["fixture_A.fasta", "fixture_B.fasta", "fixture_C.fasta", "fixture_B40.fastq", "fixture_A.fasta"]
    paths = [os.path.join(GOLD, n) for n in names]
    lst = tmp_path / "genomes.txt"
    lst.write_text("\n".join(paths) + "\n\n")
    out = str(tmp_path / "sk")
    cmd = [H.CLI, "sketch", "-f", str(lst), "-o", out, "-a", algo, "-k", str(k), "-p", str(p), "-s", "42"] + extra
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert json.load(open(out + "_files.json")) == paths
    assert open(out + "_files.json").read() == H.json_array(paths)
    params = json.load(open(out + "_parameters.json"))
    want_params = {"algorithm": algo, "k": str(k), "molecule": "nucleotide", "seed": "42"}
    if algo != "hmh":
        want_params["precision"] = str(p)
    assert params == want_params
    blob = H.zstd_read(out + "_sketches.bin")
    ib = O.image_bytes(ALGO[algo], p)
    assert len(blob) == ib * len(paths)
    for i, path in enumerate(paths):
        recs = read_fastx(path)
        seq = np.frombuffer(b"".join(recs), np.uint8)
        off = np.cumsum([0] + [len(x) for x in recs]).astype(np.uint64)
        want = O.sketch_genomes(ALGO[algo], k, p, 42, seq, off, np.array([0, len(recs)], np.uint64))[0].tobytes()
        assert blob[i * ib:(i + 1) * ib] == want, (algo, names[i])


def test_large_files_are_streamed_in_chunks(tmp_path):
    """BASELINE configs[4] shape through the CLI: files larger than --stream-mb are cut at record boundaries and
    accumulated on the device into one sketch; small neighbours keep their batch path and the file order."""
    import gzip
    import random
    rng = random.Random(2)
    big = O.synth_genome(31, 9_000_000).tobytes()
    # FASTA with one record longer than the 2 MiB chunk (line-boundary cut + overlap) and several shorter ones
    fa = b">r0\n" + b"\n".join(big[i:i + 70] for i in range(0, 5_000_000, 70)) + b"\n"
    pos = 5_000_000
    for j in range(1, 12):
        n = rng.randint(200_000, 500_000)
        fa += b">r%d some ACGT text\n" % j + b"\n".join(big[i:i + 80] for i in range(pos, pos + n, 80)) + b"\n"
        pos += n
    # one enormous single-line record
    fa1 = b">oneline\n" + big[:6_000_000] + b"\n>tail\nACGTACGTACGTACGTACGTACGTACGTACGT\n"
    reads = []
    for i in range(40_000):
        s = rng.randrange(0, 8_000_000)
        reads.append(b"@r%d\n" % i + big[s:s + 150] + b"\n+\n" + bytes(rng.choice(b"@+IIIIFF>") for _ in range(150)) + b"\n")
    fq = b"".join(reads)
    small = b">s\n" + big[100:50_100] + b"\n"
    contents = {"big.fa": fa, "small1.fa": small, "oneline.fa": fa1, "reads.fq": fq, "small2.fa": small[:20_000] + b"\n"}
    paths = []
    for name, data in contents.items():
        (tmp_path / name).write_bytes(data)
        paths.append(str(tmp_path / name))
    with gzip.open(tmp_path / "reads.fq.gz", "wb", compresslevel=1) as g:
        g.write(fq)
    paths.append(str(tmp_path / "reads.fq.gz"))
    # the same reads as MANY concatenated gzip members cut at arbitrary bytes (mid-line): with -t > 1 the members inflate in
    # parallel (host/pgzip.hpp) and must come out as one seamless stream
    cuts = sorted({0, len(fq)} | {rng.randrange(len(fq)) for _ in range(37)})
    (tmp_path / "reads_mm.fq.gz").write_bytes(b"".join(gzip.compress(fq[a:b], 1) for a, b in zip(cuts[:-1], cuts[1:])))
    paths.append(str(tmp_path / "reads_mm.fq.gz"))
    import bz2
    import lzma
    (tmp_path / "big.fa.bz2").write_bytes(bz2.compress(fa, 1))            # streamed (large once inflated)
    (tmp_path / "small1.fa.xz").write_bytes(lzma.compress(small, format=lzma.FORMAT_XZ, preset=1))   # batch path
    (tmp_path / "reads.fq.xz").write_bytes(lzma.compress(fq, format=lzma.FORMAT_XZ, preset=0))
    paths += [str(tmp_path / "big.fa.bz2"), str(tmp_path / "small1.fa.xz"), str(tmp_path / "reads.fq.xz")]
    lst = tmp_path / "l.txt"
    lst.write_text("\n".join(paths) + "\n")
    for algo, k, p in (("hmh", 16, 10), ("ull", 21, 12)):
        out = str(tmp_path / ("big_" + algo))
        workers = ["--devices", "0,0"] if algo == "ull" else []        # two GPU workers: batches finish out of order
        r = subprocess.run([H.CLI, "sketch", "-f", str(lst), "-o", out, "-a", algo, "-k", str(k), "-p", str(p),
                            "--stream-mb", "2", "--batch-mb", "1", "-t", "4"] + workers, capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        blob = H.zstd_read(out + "_sketches.bin")
        ib = O.image_bytes(ALGO[algo], p)
        assert len(blob) == ib * len(paths)
        for i, path in enumerate(paths):
            src = path.rsplit(".", 1)[0].replace("reads_mm.fq", "reads.fq") if path.endswith((".gz", ".bz2", ".xz")) else path
            recs = read_fastx(src)
            seq = np.frombuffer(b"".join(recs), np.uint8)
            off = np.cumsum([0] + [len(x) for x in recs]).astype(np.uint64)
            want = O.sketch_genomes(ALGO[algo], k, p, 42, seq, off, np.array([0, len(recs)], np.uint64), threads=8)[0].tobytes()
            assert blob[i * ib:(i + 1) * ib] == want, (algo, os.path.basename(path))


def test_cli_stops_at_malformed_fastq_records_like_needletail(tmp_path):
    """`lash sketch` validates FASTQ in its reader threads (lash_fastq_valid_prefix): structure AND sequence/quality length, so a
    malformed record ends the file's contribution exactly where needletail's iterator would end (utils.rs:457) — in the
    batch path (tail blanked inside the batch buffer) and in the streamed path (large file: streaming stops at that chunk)."""
    import random
    rng = random.Random(77)
    base = O.synth_genome(55, 3_000_000).tobytes()

    def reads(n, off):
        return [base[off + 131 * i: off + 131 * i + rng.choice([100, 150, 151])] for i in range(n)]

    def fq(rs, q=b"F"):
        return b"".join(b"@r%d\n%s\n+\n%s\n" % (i, r, q * len(r)) for i, r in enumerate(rs))
    a, b = reads(3000, 0), reads(3000, 500_000)
    files = {
        "ok.fq": fq(a) + fq(b),
        "short_quality.fq": fq(a) + b"@bad\n" + b[0] + b"\n+\n" + b"F" * (len(b[0]) - 1) + b"\n" + fq(b),
        "blank_line.fq": fq(a) + b"\n" + fq(b),
        "no_plus.fq": fq(a) + b"@bad\n" + b[0] + b"\n" + b[1] + b"\n" + fq(b),
        "first_bad.fq": b"@bad\nACGTACGTACGTACGTACGTACGT\n+\nFFF\n" + fq(a),
        "big_short_quality.fq": fq(reads(12_000, 100)) + b"@bad\n" + b[0] + b"\n+\nF\n" + fq(reads(12_000, 900_000)),   # > --stream-mb: streamed
    }
    paths = []
    for name, data in files.items():
        (tmp_path / name).write_bytes(data)
        paths.append(str(tmp_path / name))
    (tmp_path / "l.txt").write_text("\n".join(paths) + "\n")
    for algo, k, p in (("hmh", 16, 10), ("hll", 21, 12)):
        out = str(tmp_path / ("mal_" + algo))
        r = subprocess.run([H.CLI, "sketch", "-f", str(tmp_path / "l.txt"), "-o", out, "-a", algo, "-k", str(k), "-p", str(p), "--stream-mb", "2",
                            "--batch-mb", "1", "-t", "3"], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        blob = H.zstd_read(out + "_sketches.bin")
        want = O.sketch_files(ALGO[algo], k, p if algo != "hmh" else 0, 42, list(files.values()), threads=4)
        ib = want.shape[1]
        for i, name in enumerate(files):
            assert blob[i * ib:(i + 1) * ib] == want[i].tobytes(), (algo, name)
        assert blob[ib:2 * ib] != blob[:ib]                      # the truncation is real: not the same sketch as the clean file
        # the other hypothesis of SURVEY App. D's U6 (needletail's iterator goes on after an error): --layout fastq_err=skip drops
        # the malformed record and resumes at the next record — batch path, streamed path and the library's host-buffer entry
        out2 = str(tmp_path / ("skip_" + algo))
        r = subprocess.run([H.CLI, "sketch", "-f", str(tmp_path / "l.txt"), "-o", out2, "-a", algo, "-k", str(k), "-p", str(p), "--stream-mb", "2",
                            "--batch-mb", "1", "-t", "3", "--layout", "fastq_err=skip"], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        blob2 = H.zstd_read(out2 + "_sketches.bin")
        lay = O.make_layout(fastq_err="skip")
        want2 = O.sketch_files(ALGO[algo], k, p if algo != "hmh" else 0, 42, list(files.values()), threads=4, layout=lay)
        for i, name in enumerate(files):
            assert blob2[i * ib:(i + 1) * ib] == want2[i].tobytes(), (algo, name, "skip")
        assert blob2[ib:2 * ib] != blob[ib:2 * ib]
    import lash_amd
    with lash_amd.Context(0) as ctx:
        ctx.set_layout("fastq_err=skip")
        got = ctx.sketch_files_raw("hmh", 16, 0, 42, [files["no_plus.fq"], files["ok.fq"], files["blank_line.fq"]])
        want = O.sketch_files(O.HMH, 16, 0, 42, [files["no_plus.fq"], files["ok.fq"], files["blank_line.fq"]], layout=O.make_layout(fastq_err="skip"))
        assert np.array_equal(got, want)
        assert ctx.format_errors() == [0, 2]


@pytest.mark.sole
@pytest.mark.parametrize("wgs", [None, "2"])
def test_many_small_files_and_a_multi_frame_sketch_file(tmp_path, wgs, monkeypatch):
    """A collection of small genomes (one file each): the persistent small-genome kernel behind the raw-file entry, read tasks that cover runs
    of files, and — with this image's libzstd, which has no multithreading — a sketches.bin made of several zstd frames.  The decompressed
    bytes equal the oracle's images in list order and the single-thread writer's bytes; `lash dist` reads the multi-frame file."""
    import random
    if wgs:
        monkeypatch.setenv("LASH_SOLE_WGS", wgs)              # (the child processes inherit it: many genomes per workgroup)
    rng = random.Random(9)
    paths, want = [], []
    for i in range(420):                                        # 420 x 32 KiB of images: four 4 MiB pieces
        L = rng.choice([0, 9, 40, 700, 3000, 12000, 30000])
        seq = O.synth_genome(1000 + i, max(L, 1)).tobytes()[:L]
        if L > 100 and i % 5 == 0:
            seq = seq[:L // 2] + b"NNNnnacgtN" + seq[L // 2:]
        pth = tmp_path / ("s%d.fa" % i)
        if i % 3 == 0 and L > 200:                             # two records
            pth.write_bytes(b">a\n" + seq[:L // 3] + b"\n>b\n" + seq[L // 3:] + b"\n")
            recs = [seq[:L // 3], seq[L // 3:]]
        else:
            pth.write_bytes(b">a desc\n" + b"\n".join(seq[j:j + 60] for j in range(0, len(seq), 60)) + b"\n")
            recs = [seq]
        paths.append(str(pth))
        arr = np.frombuffer(b"".join(recs), np.uint8)
        off = np.cumsum([0] + [len(x) for x in recs]).astype(np.uint64)
        want.append(O.sketch_genomes(O.HMH, 16, 0, 42, arr, off, np.array([0, len(recs)], np.uint64))[0].tobytes())
    lst = tmp_path / "l.txt"
    lst.write_text("\n".join(paths) + "\n")
    outs = {}
    for tag, env in (("frames", {}), ("one", {"LASH_ZSTD_ONE_THREAD": "1"})):
        out = str(tmp_path / tag)
        r = subprocess.run([H.CLI, "sketch", "-f", str(lst), "-o", out, "-a", "hmh", "-k", "16", "-s", "42", "-t", "4"], capture_output=True, text=True,
                           env=dict(os.environ, **env))
        assert r.returncode == 0, r.stderr
        outs[tag] = H.zstd_read(out + "_sketches.bin")
    assert outs["frames"] == outs["one"] == b"".join(want)
    raw = open(str(tmp_path / "frames") + "_sketches.bin", "rb").read()
    if raw.count(b"\x28\xb5\x2f\xfd") < 2:
        pytest.skip("this libzstd compresses with its own threads: one frame")
    # `lash dist` on the multi-frame file against the single-frame one: the same table
    tabs = []
    for tag in ("frames", "one"):
        r = subprocess.run([H.CLI, "dist", "-q", tag, "-r", tag, "-o", tag + ".tsv", "-t", "2"], capture_output=True, text=True, cwd=str(tmp_path))
        assert r.returncode == 0, r.stderr
        tabs.append(open(str(tmp_path / (tag + ".tsv"))).read())
    assert tabs[0] == tabs[1] and tabs[0].count("\n") > 420
