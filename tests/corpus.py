"""Canterbury corpus lookup shared by the tests and bench.py.

The reference fetches the corpus with download-corpora.sh into corpora/{cantrbry,large,artificl}
(/root/reference/download-corpora.sh:6-7,34,45) and its benchmark cycles each file to a fixed size before
cutting it into 64 KiB arrays (/root/reference/benchmark/Main.hs:80-84).  There is no network here, so the
files are looked up under $CANTERBURY_DIR (default: <repo>/corpora); when they are absent the callers skip
with an explicit message -- nothing is substituted."""
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# BASELINE.json configs[0] (alice29) and configs[2] (the "large" corpus); the reference's own benchmark
# uses bible.txt, world192.txt and alice29.txt (benchmark/Main.hs:42-52)
FILES = ["cantrbry/alice29.txt", "large/bible.txt", "large/E.coli", "large/world192.txt"]


def corpus_dir():
    return os.environ.get("CANTERBURY_DIR") or os.path.join(ROOT, "corpora")


def find(rel):
    p = os.path.join(corpus_dir(), rel)
    return p if os.path.isfile(p) else None


def cycled(path, n_bytes):
    """The file repeated until n_bytes (benchmark/Main.hs:80-84 normalises to 10 MiB the same way)."""
    data = open(path, "rb").read()
    if not data:
        raise ValueError("empty corpus file " + path)
    reps = (n_bytes + len(data) - 1) // len(data)
    return (data * reps)[:n_bytes]
