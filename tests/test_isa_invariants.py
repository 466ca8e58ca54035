"""CPU: invariants of the SHIPPED ISA of the implicit-GEMM kernels (hipcc cross-compiles gfx950 without a GPU).

VERDICT r03 item 4 / ADVICE r03: the select-form of dvg_conv3x3_first_pair's store phase once gave run-to-run different
tiles.  r04: that form compiles to EXEC-masked basic blocks inside the MFMA-interleaved stage loop
(31 `s_and_saveexec` there); the shipped forms have none.  This test pins that property for EVERY conv_igemm2_kernel
instantiation of both builds (bf16 triples and -DDVG_BF16X3=0): inside a kernel's loops there is
  * no EXEC change (`s_*_saveexec*`, `s_or/andn2/xor_b64 exec`) and no branch on EXECZ / EXECNZ - the `gload_a` address
    selects `((okmask >> i) & 1) ? a : dvg_zero_slot` must stay v_cndmask on loop-invariant masks, the halo / padding
    stores branch-free;
  * no lane mask is PRODUCED (no VOP3 compare into an SGPR pair): every select mask is computed before the loop;
and the loops do contain the MFMAs (the parser looks at the right blocks).  tests/test_gpu_determinism.py is the dynamic
counterpart (50 launches per instantiation, three chains in flight, bit equality)."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


def kernel_bodies(asm: str):
    out, cur = {}, None
    for line in asm.split("\n"):
        m = re.match(r"^(_ZN3dvg18conv_igemm2_kernel\w+):", line)
        if m:
            cur = m.group(1)
            out[cur] = []
            continue
        if line.startswith("\t.amdhsa_kernel") or line.startswith(".Lfunc_end"):
            cur = None
        if cur:
            out[cur].append(line)
    return out


def loop_census(lines):
    """Counts over the basic blocks hipcc annotates as loop headers / `in Loop: Header=...`, per loop (keyed by the block's
    IMMEDIATE loop header), summed over the loops that contain MFMAs: the stage loops.  Loops without an MFMA (the split-K
    epilogue's wait for the other splits' partial tiles and its sum over them, which run after the last MFMA has retired) are
    counted under "other_loops" only."""
    zero = {"mfma": 0, "exec_change": 0, "exec_branch": 0, "mask_produced": 0, "cndmask": 0}
    loops = {}
    cur = None
    for ln in lines:
        m = re.match(r"^\.L(BB\d+_\d+):", ln)
        if m or ln.startswith("; %bb."):
            h = re.search(r"in Loop: Header=(BB\d+_\d+)", ln)
            if h:
                cur = h.group(1)
            elif "Loop Header" in ln and m:
                cur = m.group(1)
            else:
                cur = None
        if cur is None:
            continue
        c = loops.setdefault(cur, dict(zero))
        op = ln.strip().split(" ")[0] if ln.strip() else ""
        if op.startswith("v_mfma"):
            c["mfma"] += 1
        if "saveexec" in op or (op in ("s_or_b64", "s_andn2_b64", "s_xor_b64", "s_and_b64", "s_mov_b64") and re.search(r"\bexec\b", ln.split(",")[0])):
            c["exec_change"] += 1
        if op in ("s_cbranch_execz", "s_cbranch_execnz"):
            c["exec_branch"] += 1
        if re.match(r"v_cmpx?_\w+_e64$", op) and re.search(r"\bs\[\d+:\d+\]", ln.split(",")[0]):
            c["mask_produced"] += 1
        if op.startswith("v_cndmask"):
            c["cndmask"] += 1
    out = dict(zero)
    out["other_loops"] = 0
    for c in loops.values():
        if c["mfma"]:
            for k in zero:
                out[k] += c[k]
        else:
            out["other_loops"] += 1
    return out


@pytest.mark.parametrize("x3", [1, 0])
def test_no_exec_change_and_no_lane_mask_inside_any_stage_loop(x3, tmp_path):
    if not os.path.exists(HIPCC):
        pytest.skip("hipcc not available")
    src = os.path.join(ROOT, "dvg_amd", "csrc", "conv_igemm2.hip")
    out = str(tmp_path / "conv_igemm2.s")
    subprocess.run([HIPCC, "-O3", "-std=c++17", "--offload-arch=gfx950", f"-DDVG_BF16X3={x3}", "-I", os.path.dirname(src),
                    "-I", os.path.join(ROOT, "include"), "-S", "--cuda-device-only", "-o", out, src], check=True,
                   stderr=subprocess.DEVNULL)
    bodies = kernel_bodies(open(out).read())
    assert len(bodies) >= 10, sorted(bodies)                      # 3 + FIRST + 2 + 3 + 2 (+ the NT = 2 GEMM tile with triples)
    assert any("Lb1EEE" in k for k in bodies), "the FIRST instantiation is missing"
    bad = {}
    for name, lines in bodies.items():
        c = loop_census(lines)
        assert c["mfma"] >= 16, (name, c)
        if c["exec_change"] or c["exec_branch"] or c["mask_produced"]:
            bad[name] = c
    assert not bad, bad
    # the address-select path exists and is predication, not control flow: conv modes keep v_cndmask in their loops
    assert sum(loop_census(v)["cndmask"] for k, v in bodies.items() if "ILi3E" not in k) > 0
    # no device-scope FENCE anywhere in these kernels: on gfx950 that is a write-back + invalidate of the XCD's whole L2
    # (measured r04 with the in-kernel split-K exchange, removed in r05: the dcgan_64 chain 3.94 -> 5.59 ms)
    text = {k: "\n".join(v) for k, v in bodies.items()}
    assert not any("buffer_wbl2" in t or "buffer_inv" in t for t in text.values())
