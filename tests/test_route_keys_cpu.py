"""SEERHIP_ROUTE (csrc/route.h) is a test hook: every key the library accepts is named in a test, and the Python side reads the variable the
way the library does."""
import glob
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _keys():
    h = open(os.path.join(ROOT, "pyseer_amd", "csrc", "route.h")).read()
    m = re.search(r"static const char \*const keys\[\] = \{(.*?)nullptr\};", h, re.S)
    return re.findall(r'"([a-z0-9_]+)"', m.group(1))


def test_every_route_key_is_named_in_a_test():
    keys = _keys()
    assert len(keys) >= 30 and len(set(keys)) == len(keys)
    me = os.path.abspath(__file__)
    text = {f: open(f).read() for f in glob.glob(os.path.join(ROOT, "tests", "*.py")) if os.path.abspath(f) != me}
    missing = [k for k in keys if not any(re.search(r"(?<![A-Za-z0-9_])" + k + r"(?![A-Za-z0-9_])", t) for t in text.values())]
    assert not missing, missing


def test_python_reads_the_route_like_the_library(monkeypatch):
    from pyseer_amd import _route
    monkeypatch.delenv("SEERHIP_ROUTE", raising=False)
    monkeypatch.delenv("SEERHIP_DEBUG", raising=False)
    assert _route.route("job") is None and _route.route("job", "1") == "1" and not _route.debug("cli")
    monkeypatch.setenv("SEERHIP_ROUTE", "chord=0,job=0,reader=serial")
    monkeypatch.setenv("SEERHIP_DEBUG", "host,cli")
    assert _route.route("job") == "0" and _route.route("reader") == "serial" and _route.route("jo") is None and _route.route("dma", "1") == "1"
    assert _route.debug("cli") and _route.debug("host") and not _route.debug("firth")
    assert _route.with_route("chord=0,job=0", job=None, lanes=2) == "chord=0,lanes=2"
    # the environment variables of the package: these four and no other
    names = set()
    for pat in ("pyseer_amd/*.py", "pyseer_amd/csrc/*", "bench.py"):
        for f in glob.glob(os.path.join(ROOT, pat)):
            if os.path.isfile(f) and not f.endswith((".o", ".so")):
                names |= set(re.findall(r"SEERHIP_[A-Z_]+", open(f, errors="ignore").read()))
    names -= {"SEERHIP_H", "SEERHIP_HOST_POOL_H"}
    # (names that only occur in the comment of route.h that records what round 5 folded away)
    folded = {"SEERHIP_FIRTH_LITERAL", "SEERHIP_LMM_LIMBS", "SEERHIP_QF", "SEERHIP_READER", "SEERHIP_WAIT", "SEERHIP_FIRTH_DEBUG"}
    assert names - folded == {"SEERHIP_ROUTE", "SEERHIP_DEBUG", "SEERHIP_LIB", "SEERHIP_BENCH_CPU_EIGH"}, names
