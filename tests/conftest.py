import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(autouse=True)
def _alg_env_follows_monkeypatch(monkeypatch):
    """libalg_hip.so reads its ALG_* options once, at load (include/alg_hip.h: alg_reload_env).  Tests flip them through
    `monkeypatch.setenv / delenv`; this hook makes the library follow -- and re-reads once more after monkeypatch has restored
    the environment, so that no test leaks an option into the next."""
    from alg_amd import _lib

    def reload():
        if _lib._lib is not None or os.path.exists(_lib.LIB_PATH):
            _lib.reload_env()

    setenv, delenv = monkeypatch.setenv, monkeypatch.delenv
    touched = []

    def setenv_(name, value, *a, **k):
        setenv(name, value, *a, **k)
        if name.startswith("ALG_"):
            touched.append(name)
            reload()

    def delenv_(name, *a, **k):
        delenv(name, *a, **k)
        if name.startswith("ALG_"):
            touched.append(name)
            reload()

    monkeypatch.setenv, monkeypatch.delenv = setenv_, delenv_
    yield
    if touched:
        monkeypatch.undo()
        reload()


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session")
def device():
    import torch

    if not torch.cuda.is_available():
        pytest.skip("no GPU visible")
    return torch.device("cuda:0")


@pytest.fixture
def make_tokenizer_dir():
    """Writes a tiny T5-style tokenizer (a sentencepiece unigram model trained on the spot, saved with transformers'
    `save_pretrained`) into `<root>/<subfolder>`: the on-disk form of a checkpoint's `tokenizer/` directory."""
    spm = pytest.importorskip("sentencepiece")
    transformers = pytest.importorskip("transformers")

    def make(root, subfolder="tokenizer"):
        d = os.path.join(root, subfolder)
        os.makedirs(d, exist_ok=True)
        corpus = os.path.join(root, "corpus.txt")
        lines = ["a red double decker bus driving down a street", "a small boat drifts on the lake", "blurry static low quality",
                 "the kite flies over the dunes", "a paper boat in the rain", "people walking in a park at sunset"]
        with open(corpus, "w") as f:
            f.write("\n".join(lines * 20))
        spm.SentencePieceTrainer.train(input=corpus, model_prefix=os.path.join(d, "spiece"), vocab_size=64, model_type="unigram",
                                       pad_id=0, eos_id=1, unk_id=2, bos_id=-1, hard_vocab_limit=False, minloglevel=2)
        transformers.T5Tokenizer(vocab_file=os.path.join(d, "spiece.model"), extra_ids=0).save_pretrained(d)
        return d

    return make


@pytest.fixture(autouse=True)
def _poison_in_front_of_every_launch(request):
    """`ALG_TEST_POISON=<pattern>` (nan | big | neg | allbits | zero | eighty ...; tests/helpers/poison.py): every launch of the
    suite that goes through alg_amd._lib is preceded, on the same stream, by a kernel that rewrites all vector registers, the
    160 KiB of LDS and s[16:99] + vcc of every CU with the pattern.  A kernel that reads only what it wrote cannot tell, so the
    whole GPU suite must stay green.  Off by default (it doubles the number of launches); graph-capture tests are left alone
    (the poison launch would be captured too)."""
    pat = os.environ.get("ALG_TEST_POISON")
    if not pat or "graph" in request.node.nodeid or request.node.get_closest_marker("gpu") is None:
        yield
        return
    import torch

    from alg_amd import _lib
    from helpers import poison as P

    lib, orig = P.load(), _lib._stream

    def poisoned():
        s = orig()
        if not torch.cuda.is_current_stream_capturing():
            lib.reg_poison(P.PATTERNS[pat], P.ALL, 512, s)
        return s

    _lib._stream = poisoned
    try:
        yield
    finally:
        _lib._stream = orig
