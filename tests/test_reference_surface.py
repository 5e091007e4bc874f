"""The drop-in claim, checked against a machine-derived statement of the reference's call surface.

tests/golden/reference_surface.json is written by tests/golden/make_reference_surface.py IN THE BUILD CONTAINER from the
reference's own files (ast over scripts/Controller.py, main_solo12_control.py, LoggerControl.py, test_mpc.py, QP_WBC.py,
MPC_Wrapper.py, solo12InvKin.py; a regex over python/gepadd.cpp): for every object the hot path replaces, the attributes the
reference's callers read and write, the methods they call with their positional-argument counts, the constructors, and what
the bound C++ classes export.  Names and numbers only; the fixture travels, the reference does not.  This test asserts that
the drop-in modules provide every entry -- statically (ast / inspect over OUR modules: no GPU, no handle is created).
Entries that are deliberately not provided are listed in NOT_PROVIDED with the reason; entries of a reference caller that no
longer matches the reference's OWN class definition (scripts/test_mpc.py is stale) are recognised from the fixture's
`definitions` block, not from a hand-written list."""
import ast
import inspect
import json
import os

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(os.path.dirname(HERE), "quadruped-reactive-walking_amd")
SURFACE = json.load(open(os.path.join(HERE, "golden", "reference_surface.json")))

# what each root of the fixture is in the drop-in package: (module, class or None for a module)
TARGET = {
    "MPC_Wrapper.MPC_Wrapper (instance)": ("MPC_Wrapper", "MPC_Wrapper"),
    "QP_WBC.wbc_controller (instance)": ("QP_WBC", "wbc_controller"),
    "QP_WBC.wbc_controller (class)": ("QP_WBC", "wbc_controller"),
    "wbc_controller.invKin": ("QP_WBC", "_InvKinView"),
    "solo12InvKin.Solo12InvKin (instance)": ("solo12InvKin", "Solo12InvKin"),
    "solo12InvKin.Solo12InvKin (class)": ("solo12InvKin", "Solo12InvKin"),
    "lrw.MPC (instance)": ("libquadruped_reactive_walking", "MPC"),
    "lrw.MPC (instance, child process)": ("libquadruped_reactive_walking", "MPC"),
    "lrw.QPWBC (instance)": ("libquadruped_reactive_walking", "QPWBC"),
    "lrw.InvKin (instance)": ("libquadruped_reactive_walking", "InvKin"),
    "lqrw.Gait (instance)": ("libquadruped_reactive_walking", "Gait"),
    "lqrw.StatePlanner (instance)": ("libquadruped_reactive_walking", "StatePlanner"),
    "lqrw.FootstepPlanner (instance)": ("libquadruped_reactive_walking", "FootstepPlanner"),
    "lqrw.FootTrajectoryGenerator (instance)": ("libquadruped_reactive_walking", "FootTrajectoryGenerator"),
    "module MPC_Wrapper": ("MPC_Wrapper", None),
    "module libquadruped_reactive_walking": ("libquadruped_reactive_walking", None),
    "module libquadruped_reactive_walking (as MPC)": ("libquadruped_reactive_walking", None),
}

# (file, root, kind, name) the drop-ins deliberately do not provide, and why
NOT_PROVIDED = {
    ("scripts/QP_WBC.py", "self.invKin", "reads", "robot"):
        "a Pinocchio RobotWrapper: the reference's QP_WBC.py calls crba / rnea / frame Jacobians on it (QP_WBC.py:88-103); the "
        "drop-in QP_WBC.py replaces that file as a whole (rigid-body dynamics inside the WBC kernel), so solo12InvKin.py is never "
        "combined with the reference's QP_WBC.py",
    ("scripts/MPC_Wrapper.py", "self.mpc", "calls", "solve"):
        "the Crocoddyl MPC (mpc_type=False, scripts/crocoddyl_class): outside the accelerated path, MPC_Wrapper raises NotImplementedError",
    ("scripts/MPC_Wrapper.py", "loop_mpc", "calls", "solve"):
        "the Crocoddyl MPC in the child process: as above",
    ("scripts/test_mpc.py", "MPC_Wrapper", "calls", "Dummy"):
        "a container class of the reference's MPC_Wrapper.py for its shared-memory protocol (two fields, no behaviour): the "
        "stream-based asynchronous mode has no shared memory to unpack",
}


def _module_ast(mod):
    return ast.parse(open(os.path.join(PKG, mod + ".py")).read())


def _class_surface(mod, cls):
    """(methods -> (min, max) positional arguments, attributes assigned on self + class-level names) of OUR class, read from
    the source -- constructors need a GPU, the surface does not."""
    for node in _module_ast(mod).body:
        if isinstance(node, ast.ClassDef) and node.name == cls:
            methods, attrs = {}, set()
            for f in node.body:
                if isinstance(f, ast.FunctionDef):
                    a = f.args
                    npos = len(a.args) - 1
                    methods[f.name] = (npos - len(a.defaults), None if a.vararg else npos)
                    if any(isinstance(d, ast.Name) and d.id == "property" for d in f.decorator_list):
                        attrs.add(f.name)
                    for n in ast.walk(f):
                        if isinstance(n, ast.Attribute) and isinstance(n.ctx, ast.Store) and isinstance(n.value, ast.Name) and n.value.id == "self":
                            attrs.add(n.attr)
                        # self.a = self.b = ... = value and tuple targets
                        if isinstance(n, (ast.Tuple, ast.List)) and isinstance(n.ctx, ast.Store):
                            for e in n.elts:
                                if isinstance(e, ast.Attribute) and isinstance(e.value, ast.Name) and e.value.id == "self":
                                    attrs.add(e.attr)
                elif isinstance(f, ast.Assign):
                    attrs.update(t.id for t in f.targets if isinstance(t, ast.Name))
            return methods, attrs
    raise AssertionError("%s.%s not found" % (mod, cls))


def _module_names(mod):
    names = set()
    for node in _module_ast(mod).body:
        if isinstance(node, (ast.ClassDef, ast.FunctionDef)):
            names.add(node.name)
        elif isinstance(node, ast.Assign):
            names.update(t.id for t in node.targets if isinstance(t, ast.Name))
    return names


def _accepts(rng, n):
    lo, hi = rng
    return n >= lo and (hi is None or n <= hi)


def _reference_defines(file_of_class, cls, kind, name, npos=None):
    """Does the reference's OWN class accept this use?  (scripts/test_mpc.py calls an interface MPC_Wrapper.py no longer has.)"""
    d = SURFACE["definitions"].get(file_of_class, {}).get(cls)
    if d is None:
        return True
    if kind == "calls":
        return name in d["methods"] and (npos is None or _accepts(d["methods"][name], npos))
    return name in d["attributes"] or name in d["methods"]


REF_CLASS_OF = {"MPC_Wrapper.MPC_Wrapper (instance)": ("scripts/MPC_Wrapper.py", "MPC_Wrapper"),
                "QP_WBC.wbc_controller (instance)": ("scripts/QP_WBC.py", "wbc_controller"),
                "solo12InvKin.Solo12InvKin (instance)": ("scripts/solo12InvKin.py", "Solo12InvKin")}


def _entries():
    for rel, roots in sorted(SURFACE["python"].items()):
        for root, e in sorted(roots.items()):
            for name in e["reads"]:
                yield rel, root, e["what"], "reads", name, None
            for name in e["writes"]:
                yield rel, root, e["what"], "writes", name, None
            for name, c in sorted(e["calls"].items()):
                for n in c["positional"]:
                    yield rel, root, e["what"], "calls", name, n
            for c in e["constructed"]:
                yield rel, root, e["what"], "constructs", "__init__", c["positional"]


def test_fixture_covers_the_objects_the_hot_path_replaces():
    whats = {e["what"] for roots in SURFACE["python"].values() for e in roots.values()}
    assert whats <= set(TARGET), whats - set(TARGET)
    for need in ("MPC_Wrapper.MPC_Wrapper (instance)", "QP_WBC.wbc_controller (instance)", "lrw.MPC (instance)",
                 "lrw.QPWBC (instance)", "lrw.InvKin (instance)", "solo12InvKin.Solo12InvKin (instance)"):
        assert need in whats
    assert set(SURFACE["bindings"]["classes"]) >= {"MPC", "QPWBC", "InvKin", "Gait", "StatePlanner", "FootstepPlanner", "FootTrajectoryGenerator"}
    assert len(list(_entries())) >= 70


def test_every_use_the_reference_makes_is_provided():
    missing, stale, skipped = [], [], []
    for rel, root, what, kind, name, npos in _entries():
        if (rel, root, kind, name) in NOT_PROVIDED:
            skipped.append((rel, root, name))
            continue
        if what in REF_CLASS_OF and kind != "constructs":
            f, c = REF_CLASS_OF[what]
            if not _reference_defines(f, c, kind, name, npos):
                stale.append((rel, root, kind, name, npos))  # the reference's own class does not provide it either
                continue
        mod, cls = TARGET[what]
        if cls is None:  # a module: the name must exist, and as a class take that many constructor arguments
            if name not in _module_names(mod):
                missing.append((rel, root, kind, name, "no such name in %s" % mod))
                continue
            if kind == "calls":
                if what == "module MPC_Wrapper" and name == "MPC_Wrapper" and not _accepts(tuple(SURFACE["definitions"]["scripts/MPC_Wrapper.py"]["MPC_Wrapper"]["methods"]["__init__"]), npos):
                    stale.append((rel, root, kind, name, npos))
                    continue
                m, _ = _class_surface(mod, name)
                if "__init__" in m and not _accepts(m["__init__"], npos):
                    missing.append((rel, root, kind, name, "constructor takes %s, called with %d" % (m["__init__"], npos)))
            continue
        methods, attrs = _class_surface(mod, cls)
        if kind in ("calls", "constructs"):
            if name not in methods:
                missing.append((rel, root, kind, name, "no such method on %s.%s" % (mod, cls)))
            elif not _accepts(methods[name], npos):
                missing.append((rel, root, kind, name, "takes %s positional arguments, called with %d" % (methods[name], npos)))
        else:
            if name not in attrs and name not in methods:
                missing.append((rel, root, kind, name, "no such attribute on %s.%s" % (mod, cls)))
    assert not missing, "\n".join(map(str, missing))
    # the stale entries are exactly scripts/test_mpc.py's (an older wrapper interface: solve(k, planner))
    assert stale and {s[0] for s in stale} == {"scripts/test_mpc.py"}, stale
    assert len(skipped) == len(NOT_PROVIDED), "an entry of NOT_PROVIDED no longer occurs in the fixture: %s" % (set(NOT_PROVIDED) - {(a, b, "reads", c) for a, b, c in skipped},)


def test_bound_classes_export_what_gepadd_cpp_exports():
    """python/gepadd.cpp:22-229: every method of every bound class the hot path (and SURVEY 8(f)) replaces exists on the
    drop-in class with that many arguments, and the constructors take the bound argument counts."""
    for cname, e in SURVEY_CLASSES.items():
        methods, _ = _class_surface("libquadruped_reactive_walking", cname)
        for m, n in e["methods"].items():
            assert m in methods, (cname, m)
            if n:  # bp::args names given: that many positional arguments
                assert _accepts(methods[m], n), (cname, m, methods[m], n)
            else:  # no names in the binding: a getter, or the arity comes from the callers' entries above
                assert methods[m][0] == 0 or m in ("run", "refreshAndCompute"), (cname, m, methods[m])
        for n in e["constructors"]:
            assert _accepts(methods["__init__"], n), (cname, "__init__", methods["__init__"], n)


SURVEY_CLASSES = {k: v for k, v in SURFACE["bindings"]["classes"].items() if k != "Params"}  # Params: yaml configuration, not on the path


def test_reference_class_attributes_that_callers_can_see_exist():
    """Every attribute the reference's wbc_controller / Solo12InvKin / MPC_Wrapper classes assign on self and that ANY walked
    caller reads is covered above; on top, the whole public attribute block of wbc_controller (what LoggerControl and user
    scripts may read: QP_WBC.py:21-50) is provided, Pinocchio objects and the two bound sub-objects aside."""
    ref = set(SURFACE["definitions"]["scripts/QP_WBC.py"]["wbc_controller"]["attributes"])
    _, ours = _class_surface("QP_WBC", "wbc_controller")
    assert ref - ours <= {"robot", "box_qp"}, ref - ours  # the Pinocchio RobotWrapper and lrw.QPWBC live inside the fused kernel
    ref = set(SURFACE["definitions"]["scripts/solo12InvKin.py"]["Solo12InvKin"]["attributes"])
    _, ours = _class_surface("solo12InvKin", "Solo12InvKin")
    assert ref - ours <= {"robot"}, ref - ours
    ref = set(SURFACE["definitions"]["scripts/MPC_Wrapper.py"]["MPC_Wrapper"]["attributes"])
    _, ours = _class_surface("MPC_Wrapper", "MPC_Wrapper")
    # the shared-memory plumbing of the child process (Value / Array objects) has no counterpart: streams and events instead
    assert ref - ours <= {"newData", "dataIn", "dataOut", "fsteps_future", "running"}, ref - ours


def test_fixture_is_what_the_generator_produces():
    """Build container only (skips where /root/reference is absent, e.g. on the GPU box): the committed fixture is exactly what
    tests/golden/make_reference_surface.py derives from the reference's files today -- nobody edited it by hand."""
    import importlib.util

    ref = "/root/reference"
    if not os.path.isdir(os.path.join(ref, "scripts")):
        pytest.skip("the reference is not present here")
    spec = importlib.util.spec_from_file_location("make_reference_surface", os.path.join(HERE, "golden", "make_reference_surface.py"))
    gen = importlib.util.module_from_spec(spec)
    argv, sys_mod = None, __import__("sys")
    argv, sys_mod.argv = sys_mod.argv, ["make_reference_surface.py", ref]
    try:
        spec.loader.exec_module(gen)
    finally:
        sys_mod.argv = argv
    fresh = {"generated_by": "tests/golden/make_reference_surface.py", "reference": "paLeziart/quadruped-reactive-walking", "python": {},
             "bindings": {"file": "python/gepadd.cpp", "classes": gen.walk_bindings(os.path.join(ref, "python", "gepadd.cpp"))}}
    for rel, roots in gen.PY_FILES.items():
        fresh["python"][rel] = gen.walk_python(os.path.join(ref, rel), roots)
    fresh["definitions"] = {rel: gen.walk_definitions(os.path.join(ref, rel), names) for rel, names in gen.DEFINED.items()}
    assert json.loads(json.dumps(fresh, sort_keys=True)) == SURFACE
