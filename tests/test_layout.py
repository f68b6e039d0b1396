"""The oracle is test infrastructure: nothing in the product imports, links or calls it; bench.py touches it only in
its cpu_baseline leg.  CPU only."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _files(top, exts):
    for d, _, fs in os.walk(os.path.join(ROOT, top)):
        for f in fs:
            if f.endswith(exts):
                yield os.path.join(d, f)


def test_product_never_touches_the_oracle():
    for path in list(_files("mpc_amd", (".py", ".h", ".hip", ".cpp", ".hpp", "Makefile"))) + list(_files("include", (".h",))):
        text = open(path, errors="replace").read()
        assert not re.search(r"^\s*(import|from)\s+oracle\b", text, flags=re.M), path
        assert "mpc_oracle" not in text and "libmpc_oracle" not in text and "orc_" not in text, path


def test_bench_uses_the_oracle_only_for_the_cpu_baseline():
    text = open(os.path.join(ROOT, "bench.py")).read()
    body = text.split("def cpu_baseline", 1)[1].split("\ndef main", 1)[0]
    rest = text.replace(body, "")
    assert "import oracle" in body and "import oracle" not in rest and "oracle_args" not in rest


def test_nothing_reads_the_reference_at_run_time():
    for path in list(_files("mpc_amd", (".py",))) + list(_files("tests", (".py",))) + [os.path.join(ROOT, "bench.py"), os.path.join(ROOT, "__graft_entry__.py")]:
        if path.endswith("test_layout.py"):
            continue
        assert "/root/reference" not in open(path).read(), path
