"""Minimal FASTA/FASTQ record reader for tests (needletail semantics):
'>' => FASTA, multi-line, CR/LF stripped; '@' => 4-line FASTQ.  Returns list of sequence bytes."""


def read_fastx(path):
    data = open(path, "rb").read()
    if not data:
        return []
    recs = []
    if data[:1] == b">":
        cur = None
        for line in data.split(b"\n"):
            line = line.rstrip(b"\r")
            if line.startswith(b">"):
                if cur is not None:
                    recs.append(b"".join(cur))
                cur = []
            elif cur is not None:
                cur.append(line)
        if cur is not None:
            recs.append(b"".join(cur))
    elif data[:1] == b"@":
        lines = data.split(b"\n")
        for i in range(0, len(lines) - 1, 4):
            if lines[i].startswith(b"@"):
                recs.append(lines[i + 1].rstrip(b"\r"))
    else:
        raise ValueError("not FASTA/FASTQ")
    return recs
