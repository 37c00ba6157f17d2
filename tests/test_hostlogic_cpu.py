"""CPU tests of the host-side logic: INT tables of the product package against
the reference's tables (golden), state-dict contract, C-ABI symbol export."""
import ctypes
import os
import re

import pytest
import torch

from conftest import ROOT


def test_int_tables_match_reference(golden):
    from mvlt_amd import indexing as I
    g = golden("int_tables")
    assert torch.equal(I.relative_position_index(7), g["relative_position_index"])
    for H in (56, 28, 14):
        m = I.shift_attn_mask(H, H, 7, 3)
        assert torch.equal((m != 0).to(torch.int8), g[f"attn_mask_{H}"]) and m.min().item() == -100.0
        assert torch.equal(I.window_token_map(H, H, 7, 0), g[f"winmap_noshift_{H}"])
        assert torch.equal(I.window_token_map(H, H, 7, 3), g[f"winmap_shift_{H}"])
        assert torch.equal(I.patch_merge_map(H, H), g[f"mergemap_{H}"])
        w2n, n2w = I.batched_window_maps(2, H, H, 7, 3, torch.device("cpu"))
        assert torch.equal(w2n[:H * H].long(), g[f"winmap_shift_{H}"])
        assert torch.equal(w2n.long()[n2w.long()], torch.arange(2 * H * H))


@pytest.mark.parametrize("name,cls", [("pretrain", "MVLBertForPretraining"), ("vqa", "MVLBertForVQA")])
def test_state_dict_contract(specs, name, cls):
    """Same keys, shapes, dtypes and order as the reference model's state_dict()."""
    import mvlt_amd as M
    cfg = M.MVLBertPretrainConfig() if name == "pretrain" else M.MVLBertConfigforVQA()
    model = getattr(M, cls)(cfg)
    ref = [(k, tuple(s), d) for k, s, d in specs[name]]
    mine = [(k, tuple(v.shape), str(v.dtype).replace("torch.", "")) for k, v in model.state_dict().items()]
    assert [k for k, _, _ in mine] == [k for k, _, _ in ref]
    assert mine == ref
    if name == "pretrain":
        assert sum(p.numel() for p in model.parameters()) == 208853340


def test_tiny_caption_contract(specs):
    import mvlt_amd as M
    cfg = M.MVLBertConfigForImageCaption(hidden_size=256, num_hidden_layers=2, num_attention_heads=4,
                                         intermediate_size=1024, vocab_size=3000)
    cfg.swin.update(embed_dim=32, depths=[2, 2, 2, 2], num_heads=[1, 2, 4, 8], drop_path_rate=0.2)
    model = M.MVLBertForImageCaption(cfg)
    ref = [(k, tuple(s)) for k, s, d in specs["tiny_caption"]]
    assert [(k, tuple(v.shape)) for k, v in model.state_dict().items()] == ref


def test_swin_flops_match_reference_formula():
    import mvlt_amd as M
    sw = M.SwinTransformer(embed_dim=96, depths=[2, 2, 18, 2], num_heads=[3, 6, 12, 24], drop_path_rate=0.3)
    assert abs(sw.flops() / 1e9 - 8.746) < 0.002          # 8.746 GMAC / image


def test_cabi_exports_every_declared_symbol():
    """libmvlt_hip.so loads on a GPU-less host and exports every function
    include/mvlt_hip.h declares (no compute calls here)."""
    from mvlt_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "mvlt_hip.h")).read()
    declared = set(re.findall(r"\b(mvlt_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(_lib.SYMBOLS), declared ^ set(_lib.SYMBOLS)
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), name
    L = _lib.lib()
    assert L.mvlt_arch() == b"gfx950"


def test_cabi_version_and_struct_sizes_agree_everywhere():
    """VERDICT r2 item 8 / ADVICE r2: one ABI constant in the header, returned by the library, compiled into the
    torch extension and hard-coded in the ctypes mirror; every struct mirror has the size the library was compiled
    with (mvlt_sizeof), and the extension's own sizeof()s agree too -- a stale build of any one piece fails here."""
    from mvlt_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "mvlt_hip.h")).read()
    macro = int(re.search(r"#define\s+MVLT_ABI_VERSION\s+(\d+)", hdr).group(1))
    L = _lib.lib()
    assert macro == _lib.ABI_VERSION == L.mvlt_version()
    ids = re.search(r"enum \{ (MVLT_STRUCT_GEMM.*?)\};", hdr, re.S).group(1)
    names = [n.split("=")[0].strip() for n in ids.split(",")]
    assert names[-1] == "MVLT_STRUCT_COUNT" and len(names) - 1 == len(_lib.STRUCTS)
    typedefs = re.findall(r"typedef struct (Mvlt\w+)", hdr)
    assert sorted(typedefs) == sorted(s.__name__ for s in _lib.STRUCTS), "a struct of the header has no ctypes mirror"
    for sid, st in enumerate(_lib.STRUCTS):
        assert L.mvlt_sizeof(sid) == ctypes.sizeof(st) > 0, st.__name__
    assert L.mvlt_sizeof(len(_lib.STRUCTS)) == 0 and L.mvlt_sizeof(-1) == 0
    # the torch extension carries its own compile-time copies (no GPU needed to import it)
    import importlib.util
    spec = importlib.util.spec_from_file_location(
        "_mvlt_host", os.path.join(os.path.dirname(_lib.LIB_PATH), "_mvlt_host.so"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    assert mod.abi_version() == macro
    assert list(mod.struct_sizes()) == [ctypes.sizeof(s) for s in _lib.STRUCTS]


def _integration_example():
    md = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    block = md.split("<!-- cabi-example-begin")[1].split("<!-- cabi-example-end -->")[0]
    return block.split("```python\n")[1].split("```")[0]


def test_integration_md_example_matches_the_header():
    """The binding snippet a maintainer would copy from INTEGRATION.md is executed (module level: CDLL load, ABI and
    sizeof checks, no GPU call) and its struct mirror must equal the package's own, field for field."""
    from mvlt_amd import _lib
    ns = {}
    os.environ["MVLT_REPO"] = ROOT
    exec(compile(_integration_example(), "INTEGRATION.md", "exec"), ns)
    mine = [(n, t) for n, t in _lib.MvltLayerNorm._fields_]
    theirs = [(n, t) for n, t in ns["MvltLayerNorm"]._fields_]
    assert [n for n, _ in mine] == [n for n, _ in theirs]
    assert all(ctypes.sizeof(a) == ctypes.sizeof(b) for (_, a), (_, b) in zip(mine, theirs))
    assert ns["MVLT_ABI_VERSION"] == _lib.ABI_VERSION


def test_no_cpu_fallback():
    import mvlt_amd as M
    sw = M.SwinTransformer(embed_dim=32, depths=[2, 2, 2, 2], num_heads=[1, 2, 4, 8])
    with pytest.raises(RuntimeError, match="GPU only"):
        sw(torch.zeros(1, 3, 224, 224))


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "medical-vision-langauge-transformer_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in src and "from oracle" not in src and "mvlt_oracle" not in src, f


def _toy_arena():
    import torch.nn as nn
    from mvlt_amd.arena import Arena

    class Toy(nn.Module):
        def __init__(self):
            super().__init__()
            self.a, self.common, self.b = nn.Linear(4, 4), nn.Linear(4, 4), nn.Linear(4, 4)
    m = Toy()
    return m, Arena(m, torch.float32, allow_cpu=True)


def test_publish_grads_rehands_views_after_zero_grad_with_a_changing_marked_set():
    """stock optimizer zero_grad() drops every p.grad view; when the marked set then changes
    (seq2seq/bidir coin flip swaps the MLM head) the parameters common to both steps must get their view back."""
    m, ar = _toy_arena()
    ar.begin_backward(); ar.mark(m.a.weight, m.common.weight); ar.publish_grads()
    assert m.a.weight.grad is not None and m.common.weight.grad is not None and m.b.weight.grad is None
    m.zero_grad(set_to_none=True)
    ar.begin_backward(); ar.mark(m.b.weight, m.common.weight); ar.publish_grads()
    assert m.common.weight.grad is not None and m.b.weight.grad is not None and m.a.weight.grad is None
    assert m.common.weight.grad.data_ptr() == ar.grad_view(m.common.weight).data_ptr()


def test_unconsumed_gradients_accumulate_and_consumed_ones_do_not():
    m, ar = _toy_arena()
    ar.begin_backward(); ar.grad_view(m.a.weight).fill_(1.0); ar.grad_view(m.common.weight).fill_(2.0)
    ar.mark(m.a.weight, m.common.weight); ar.publish_grads()
    # second pass without clearing: a different set; common accumulates, a keeps its value, b is new
    ar.begin_backward(); ar.grad_view(m.b.weight).fill_(5.0); ar.grad_view(m.common.weight).fill_(3.0)
    ar.mark(m.b.weight, m.common.weight); ar.publish_grads()
    assert float(m.common.weight.grad[0, 0]) == 5.0 and float(m.a.weight.grad[0, 0]) == 1.0
    assert float(m.b.weight.grad[0, 0]) == 5.0 and ar.has_grad[id(m.a.weight)]
    # cleared in between (zero_grad set_to_none) -> plain overwrite
    m.zero_grad(set_to_none=True)
    ar.begin_backward(); ar.grad_view(m.common.weight).fill_(7.0); ar.mark(m.common.weight); ar.publish_grads()
    assert float(m.common.weight.grad[0, 0]) == 7.0 and m.a.weight.grad is None
    # consumed by the fused optimizer -> overwrite although the views are still attached
    ar.note_grads_consumed()
    ar.begin_backward(); ar.grad_view(m.common.weight).fill_(1.5); ar.mark(m.common.weight); ar.publish_grads()
    assert float(m.common.weight.grad[0, 0]) == 1.5


def test_hardware_queue_default_is_set_on_import():
    """mvlt_amd sets GPU_MAX_HW_QUEUES=8 on import unless the user chose a value (profiles/r5_ddp_one_rank.md: with the default 4
    hardware queues the step's two streams and RCCL's share queues and serialise); a process that had initialised HIP first is
    flagged so that GradReducer can warn."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import os, sys; sys.path.insert(0, %r); os.environ.pop('GPU_MAX_HW_QUEUES', None); import mvlt_amd; "
            "print(os.environ['GPU_MAX_HW_QUEUES'], mvlt_amd.HWQ_SET_LATE)") % root
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=dict(os.environ))
    assert out.stdout.split() == ["8", "False"], (out.stdout, out.stderr[-500:])
    code2 = code.replace("os.environ.pop('GPU_MAX_HW_QUEUES', None)", "os.environ['GPU_MAX_HW_QUEUES'] = '16'")
    out = subprocess.run([sys.executable, "-c", code2], capture_output=True, text=True, env=dict(os.environ))
    assert out.stdout.split() == ["16", "False"], (out.stdout, out.stderr[-500:])
