"""CPU tier: every call the REFERENCE's own host files make into the two modules this package replaces -- `gaussian`
(the pybind11 extension, src/bindings.cpp:21-50) and `renderer` (renderer.py) -- binds to the replacement's signature.

The reference's Python cannot run here (CUDA-only device, third-party modules absent) and does not travel to the GPU box,
so "renderer.py / splatter.py / train.py work unchanged on top of this package"  is checked
where it can be: the call sites are taken from the reference's sources with `ast` -- number of positional arguments and
keyword names of each `gaussian.f(...)` / `renderer.f(...)` / `f(...)` imported from renderer -- and bound against the
shim's functions with inspect.signature().bind().  What the calls COMPUTE is the business of the GPU parity tests
(tests/test_gpu_compat_pipeline.py drives the same sequence with real tensors).  Skipped when /root/reference is absent."""
import ast
import inspect
import os

import pytest

REF = "/root/reference"
pytestmark = pytest.mark.skipif(not os.path.isdir(REF), reason="reference sources not present")


def module_calls(path, module_names, imported_from):
    """(function name, n positional args, keyword names, line) of every call into one of `module_names` (attribute
    calls `mod.f(...)`) or of a name imported `from <imported_from> import f`."""
    tree = ast.parse(open(path).read(), filename=path)
    direct = set()
    for node in ast.walk(tree):
        if isinstance(node, ast.ImportFrom) and node.module in imported_from:
            direct.update(a.asname or a.name for a in node.names)
    out = []
    for node in ast.walk(tree):
        if not isinstance(node, ast.Call):
            continue
        fn = node.func
        name = mod = None
        if isinstance(fn, ast.Attribute) and isinstance(fn.value, ast.Name) and fn.value.id in module_names:
            mod, name = fn.value.id, fn.attr
        elif isinstance(fn, ast.Name) and fn.id in direct:
            mod, name = next(iter(imported_from)), fn.id
        if name is None or any(isinstance(a, ast.Starred) for a in node.args) or any(k.arg is None for k in node.keywords):
            continue
        out.append((mod, name, len(node.args), tuple(k.arg for k in node.keywords), node.lineno))
    return out


def test_every_reference_call_into_gaussian_and_renderer_binds_to_the_replacement():
    import gaussian
    import renderer

    targets = {"gaussian": gaussian, "renderer": renderer}
    seen = []
    for fname in ("renderer.py", "splatter.py", "train.py", "visergui.py"):
        path = os.path.join(REF, fname)
        if not os.path.exists(path):
            continue
        for mod, name, npos, kws, line in module_calls(path, {"gaussian", "renderer"}, {"renderer"}):
            obj = getattr(targets[mod], name, None)
            assert obj is not None, f"{fname}:{line} calls {mod}.{name}, which the replacement does not have"
            if inspect.isclass(obj):  # gaussian.Tiles() / gaussian.Gaussian3ds(): default-constructible bags
                assert npos == 0 and not kws, f"{fname}:{line}"
                obj()
            else:
                sig = inspect.signature(obj)
                try:
                    sig.bind(*([None] * npos), **{k: None for k in kws})
                except TypeError as e:
                    raise AssertionError(f"{fname}:{line} {mod}.{name} with {npos} positional + {kws}: {e}") from e
            seen.append((fname, line, f"{mod}.{name}"))
    names = {s[2] for s in seen}
    # the call sites SURVEY.md section 8b lists (renderer.py:29,64,109,116,128,146; splatter.py:62,238,295,572,597 and
    # the renderer functions splatter.py imports)
    for must in ("gaussian.draw", "gaussian.draw_backward", "gaussian.world2camera", "gaussian.world2camera_backward",
                 "gaussian.global_culling", "gaussian.global_culling_backward", "gaussian.calc_tile_list",
                 "gaussian.gather_gaussians", "gaussian.jacobian", "gaussian.Tiles", "gaussian.Gaussian3ds",
                 "renderer.draw", "renderer.global_culling"):
        assert must in names, (must, sorted(names))


def test_attribute_bags_accept_what_the_reference_stores_in_them():
    """splatter.py:61-67 (Gaussian3ds._tocpp) and :295-300 (Tiles._tocpp) assign these attributes on the C++ objects."""
    import gaussian

    tree = ast.parse(open(os.path.join(REF, "splatter.py")).read())
    stored = set()
    for node in ast.walk(tree):
        if isinstance(node, ast.Assign):
            for t in node.targets:
                if isinstance(t, ast.Attribute) and isinstance(t.value, ast.Name) and t.value.id in ("_cobj", "_tile"):
                    stored.add((t.value.id, t.attr))
    assert stored, "the reference's _tocpp assignments were not found"
    g, t = gaussian.Gaussian3ds(), gaussian.Tiles()
    for owner, attr in stored:
        assert hasattr(g if owner == "_cobj" else t, attr), (owner, attr)
