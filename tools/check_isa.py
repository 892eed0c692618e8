"""Build-time checks on the generated gfx950 ISA (no GPU needed: hipcc cross-compiles).

check_chase_publish(): in `sb2st_chase` every store of a sweep's progress counter must be preceded by an explicit
`s_waitcnt vmcnt(0)` with no band access (any load, any sc1 / buffer store) in between -- the band stores of the step are write-through
(sc1) stores of the same wave and have to be complete before the counter a wave on another XCD polls moves (a workgroup-scope
release fence compiles to `lgkmcnt(0)` only).  Used by tests/test_isa_checks.py; `python tools/check_isa.py` prints the result.
"""
import os
import re
import subprocess
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "fidelityfusion_amd", "csrc")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


def device_asm(src, extra=()):
    """gfx950 assembly text of one translation unit of the library."""
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "k.s")
        subprocess.check_call([HIPCC, "-O3", "-std=c++17", "--offload-arch=gfx950", "-S", "--cuda-device-only",
                               "-Wno-unused-value", "-Wno-unused-result", "-Wno-unused-command-line-argument", *extra,
                               "-o", out, os.path.join(CSRC, src)])
        return open(out).read()


def function_body(asm, mangled_substr):
    """Instruction lines of the first function whose label contains `mangled_substr`."""
    lines = asm.splitlines()
    start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\w*%s\w*:" % re.escape(mangled_substr), l))
    end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
    return lines[start:end + 1]


def check_chase_publish(asm=None):
    """both instantiations: the chip-wide form (ILb0E) and the XCD-local one (ILb1E: plain band stores, same publish protocol)"""
    asm = asm or device_asm("sb2st.hip")
    return [_check_chase_publish_one(asm, "sb2st_chaseILb0E"), _check_chase_publish_one(asm, "sb2st_chaseILb1E")]


def _check_chase_publish_one(asm, name):
    body = function_body(asm, name)
    labels = {l.split(":")[0].strip(): i for i, l in enumerate(body) if re.match(r"^\.LBB\d+_\d+:", l)}
    waits = [i for i, l in enumerate(body) if l.strip() == "s_waitcnt vmcnt(0)" and i > 0 and "ASMSTART" in body[i - 1]]
    counter = re.compile(r"global_store_dword\s+v\d+, v\d+, s\[\d+:\d+\](?: offset:-?\d+)? sc1")

    def band_access(ins):
        # band traffic is device-scope (sc1) -- global_* in the general body, buffer_* in the interior-step loop; a plain store to the
        # reflector / tau arrays (read only after the kernel) may be scheduled in between
        return bool(re.match(r"(global|buffer|flat)_(load|atomic)", ins) or ins.startswith("buffer_") or
                    (re.match(r"(global|flat)_store", ins) and " sc1" in ins))
    found = []
    for w in waits:
        hit = None
        for j in range(w + 1, min(w + 60, len(body))):
            ins = body[j].strip()
            if counter.match(ins):
                hit = j
                break
            m = re.match(r"s_cbranch_execnz (\.LBB\d+_\d+)", ins)     # `if (lane == 0) store` laid out of line
            if m and m.group(1) in labels:
                blk = [l.strip() for l in body[labels[m.group(1)] + 1: labels[m.group(1)] + 14]]
                for k, b_ in enumerate(blk):
                    if counter.match(b_):
                        hit = labels[m.group(1)] + 1 + k
                        break
                    if band_access(b_) or b_.startswith("s_branch") or b_.startswith("s_cbranch"):
                        break
                if hit is not None:
                    break
                continue
            if band_access(ins):
                raise AssertionError("band access between the drain and the counter store: " + ins)
        if hit is not None:
            found.append((w, hit))
    assert len(found) >= 3, "expected a drained counter store per interior step, per general step and at the end of a sweep, found %d" % len(found)
    return found


def check_diag_barriers(asm=None):
    """In `ffgp_potrf_diag128_v3` the role hand-out publishes flag words with inline-asm LDS stores and then meets a workgroup barrier:
    the compiler's wait-count pass does not see inside inline asm, so every `s_barrier` must be preceded by an explicit
    `s_waitcnt lgkmcnt(0)` (round 4: without it a wave passed the barrier with its store in flight, the waves disagreed about their
    roles under load and a hand-off timed out)."""
    body = function_body(asm or device_asm("potrf.hip"), "ffgp_potrf_diag128_v3")
    bars = [i for i, l in enumerate(body) if l.strip() == "s_barrier"]
    assert len(bars) == 2, "expected the two barriers of the role hand-out, found %d" % len(bars)
    for b in bars:
        prev = [l.strip() for l in body[max(0, b - 16):b] if l.strip() and not l.strip().startswith(";")]
        drains = [j for j, l in enumerate(prev) if re.match(r"s_waitcnt .*lgkmcnt\(0\)", l)]
        assert drains, "no lgkmcnt(0) drain in front of s_barrier: %r" % prev
        after = prev[drains[-1] + 1:]
        assert not any(l.startswith("ds_") for l in after), "an LDS instruction between the drain and the barrier: %r" % after
    return bars


if __name__ == "__main__":
    print("ffgp_potrf_diag128_v3: barriers behind an LDS drain:", check_diag_barriers())
    print("sb2st_chase: counter stores behind an explicit vmcnt(0):", check_chase_publish())
