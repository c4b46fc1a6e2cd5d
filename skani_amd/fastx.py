"""Host-side FASTA ingest with the record semantics the reference gets from needletail
(file_io.rs:158-181): record id = whole header line without '>', sequence = lines joined with
line breaks removed, no case change; gzip sniffed by magic.  FASTQ/bz2/xz are out of scope.
"""
import gzip
import io


def _open(path):
    f = open(path, "rb")
    magic = f.read(2)
    f.seek(0)
    if magic == b"\x1f\x8b":
        return gzip.open(f, "rb")
    return f


def read_fasta(path):
    """Yield (name: str, seq: bytes) per record."""
    with _open(path) as f:
        data = f.read()
    if not data.strip():
        return
    if not data.lstrip().startswith(b">"):
        raise ValueError(f"{path} is not a FASTA file")
    for block in data.split(b">")[1:]:
        nl = block.find(b"\n")
        if nl < 0:
            header, body = block, b""
        else:
            header, body = block[:nl], block[nl + 1:]
        seq = body.replace(b"\n", b"").replace(b"\r", b"")
        yield header.rstrip(b"\r").decode("utf-8", "replace"), seq
