"""Two regressions of round 5 that only show on a GPU box, guarded on the CPU:
  * fastvideo_amd/_build.py — a library whose build-flag stamp differs is stale even when it is newer than every source (the packed-fp32
    fence is a flag-only change);
  * bench.py — the untimed power / clock repeat of the K steps must run on EVERY rank (a sequence-parallel forward holds collectives: with the
    repeat under `if rank == 0` the two-rank bench deadlocked, 40 GPU-minutes of round 5), only the sampler is rank 0's."""
import ast
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_flag_stamp_makes_a_library_stale(tmp_path, monkeypatch):
    from fastvideo_amd import _build
    lib = tmp_path / "libfvk_amd.so"
    lib.write_bytes(b"x")
    monkeypatch.setattr(_build, "LIB", str(lib))
    future = os.path.getmtime(str(lib)) + 10_000
    os.utime(str(lib), (future, future))                      # newer than every source
    assert _build.needs_build() is True                       # no stamp at all
    (tmp_path / "libfvk_amd.so.flags").write_text(_build._flag_stamp(False) + "\n")
    assert _build.needs_build() is False
    monkeypatch.setattr(_build, "FLAGS", _build.FLAGS + ["-DSOMETHING_NEW=1"])
    assert _build.needs_build() is True                       # same sources, other flags
    monkeypatch.undo()
    monkeypatch.setattr(_build, "LIB", str(lib))
    monkeypatch.setattr(_build, "NO_PACKED_FP32", ())         # the fence dropped: stale again
    assert _build.needs_build() is True


def _calls_model(node):
    return any(isinstance(n, ast.Call) and isinstance(n.func, ast.Name) and n.func.id == "model" for n in ast.walk(node))


def test_bench_power_repeat_runs_on_every_rank():
    tree = ast.parse(open(os.path.join(ROOT, "bench.py")).read())
    main = next(n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == "main")
    blocks = [n for n in ast.walk(main) if isinstance(n, ast.If) and "no_power_trace" in ast.unparse(n.test)]
    assert len(blocks) == 1, "bench.py: expected one `if not args.no_power_trace:` block in main()"
    body = blocks[0].body
    loops = [n for n in body if isinstance(n, ast.For) and _calls_model(n)]
    assert len(loops) == 1, "the repeat of the K steps must sit directly in the power block (every rank runs it)"
    for n in body:
        if isinstance(n, ast.If) and "rank" in ast.unparse(n.test):
            assert not _calls_model(n), "a model forward under a rank test inside the power block: the other ranks would wait in its collectives forever"
    # and the timed region's host-clock stamps exist whatever the flags
    src = ast.unparse(main)
    assert src.index("wall0 = time.time()") < src.index("no_power_trace") and "timed_region_unix" in src
