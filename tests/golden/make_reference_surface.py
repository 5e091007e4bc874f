"""Build container only: derive, from the reference's OWN files, the call surface its control loop uses on the hot path's
objects, and write it as tests/golden/reference_surface.json (names and numbers only -- no source text travels).

  python tests/golden/make_reference_surface.py [/root/reference]

What is walked:
  scripts/Controller.py, scripts/QP_WBC.py, scripts/MPC_Wrapper.py, scripts/solo12InvKin.py   with `ast`:
      for every attribute chain rooted at one of the objects the hot path replaces (self.mpc_wrapper, self.myController,
      self.invKin, self.box_qp, self.mpc, the module aliases MPC_Wrapper / lqrw / lrw / MPC and the imported names
      wbc_controller / Solo12InvKin): the attributes READ, the attributes WRITTEN, the METHODS called with their positional
      argument counts and keyword names, and the constructors called with their argument counts;
  python/gepadd.cpp   with a regex:
      for every bound class: its name, its constructors' argument counts (bp::init<...>) and its methods with the number of
      bp::args names (0 where the binding names none: getters);
  the class definitions of scripts/MPC_Wrapper.py, scripts/QP_WBC.py, scripts/solo12InvKin.py   with `ast`:
      per method the positional-argument range, and the attributes the class assigns on self.
tests/test_reference_surface.py (CPU) asserts that the drop-in modules provide every entry with a compatible arity."""
import ast
import json
import os
import re
import sys

REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_surface.json")

# (file, roots): a root is a dotted prefix of an attribute chain; what follows the root is recorded
PY_FILES = {
    "scripts/Controller.py": {"self.mpc_wrapper": "MPC_Wrapper.MPC_Wrapper (instance)", "self.myController": "QP_WBC.wbc_controller (instance)",
                              "MPC_Wrapper": "module MPC_Wrapper", "lqrw": "module libquadruped_reactive_walking",
                              "wbc_controller": "QP_WBC.wbc_controller (class)",
                              "self.gait": "lqrw.Gait (instance)", "self.statePlanner": "lqrw.StatePlanner (instance)",
                              "self.footstepPlanner": "lqrw.FootstepPlanner (instance)",
                              "self.footTrajectoryGenerator": "lqrw.FootTrajectoryGenerator (instance)"},
    "scripts/main_solo12_control.py": {"controller.mpc_wrapper": "MPC_Wrapper.MPC_Wrapper (instance)",
                                       "controller.myController": "QP_WBC.wbc_controller (instance)"},
    "scripts/LoggerControl.py": {"wbc": "QP_WBC.wbc_controller (instance)", "wbc.invKin": "wbc_controller.invKin"},
    "scripts/test_mpc.py": {"self.mpc_wrapper": "MPC_Wrapper.MPC_Wrapper (instance)", "MPC_Wrapper": "module MPC_Wrapper"},
    "scripts/QP_WBC.py": {"self.invKin": "solo12InvKin.Solo12InvKin (instance)", "self.box_qp": "lrw.QPWBC (instance)",
                          "lrw": "module libquadruped_reactive_walking", "Solo12InvKin": "solo12InvKin.Solo12InvKin (class)"},
    "scripts/MPC_Wrapper.py": {"self.mpc": "lrw.MPC (instance)", "MPC": "module libquadruped_reactive_walking (as MPC)",
                               "loop_mpc": "lrw.MPC (instance, child process)"},
    "scripts/solo12InvKin.py": {"self.InvKinCpp": "lrw.InvKin (instance)", "lrw": "module libquadruped_reactive_walking"},
}


def dotted(node):
    """'a.b.c' of an attribute chain of Names / Attributes, None for anything else."""
    parts = []
    while isinstance(node, ast.Attribute):
        parts.append(node.attr)
        node = node.value
    if isinstance(node, ast.Name):
        parts.append(node.id)
        return ".".join(reversed(parts))
    return None


def walk_python(path, roots):
    tree = ast.parse(open(path).read())
    for parent in ast.walk(tree):
        for child in ast.iter_child_nodes(parent):
            child._parent = parent
    out = {r: {"what": w, "reads": set(), "writes": set(), "calls": {}, "constructed": []} for r, w in roots.items()}

    def match(d):
        best = None
        for r in roots:
            if d == r or d.startswith(r + "."):
                if best is None or len(r) > len(best):
                    best = r
        return best

    for node in ast.walk(tree):
        if isinstance(node, ast.Call):
            d = dotted(node.func)
            if d is None:
                continue
            r = match(d)
            if r is None:
                continue
            rest = d[len(r):].lstrip(".")
            npos, kws = len(node.args), sorted(k.arg for k in node.keywords if k.arg)
            if rest == "":  # the root itself is called: a constructor (imported class name)
                out[r]["constructed"].append({"positional": npos, "keywords": kws})
            else:
                first = rest.split(".")[0]
                if "." in rest:  # call on a sub-object (e.g. self.myController.invKin.foo()): the sub-object is read
                    out[r]["reads"].add(first)
                    continue
                e = out[r]["calls"].setdefault(first, {"positional": set(), "keywords": set()})
                e["positional"].add(npos)
                e["keywords"].update(kws)
        elif isinstance(node, ast.Attribute):
            par = getattr(node, "_parent", None)
            if isinstance(par, ast.Attribute) and par.value is node:
                continue  # an inner link of a longer chain: the outermost node reports it
            if isinstance(par, ast.Call) and par.func is node:
                continue  # reported as a call
            d = dotted(node)
            if d is None:
                continue
            r = match(d)
            if r is None or d == r:
                continue
            first = d[len(r):].lstrip(".").split(".")[0]
            store = isinstance(node.ctx, ast.Store) and d == r + "." + first
            # a subscript assignment (obj.attr[...] = v) reads the attribute object and mutates it: the attribute must exist
            out[r]["writes" if store else "reads"].add(first)
    res = {}
    for r, e in out.items():
        if not (e["reads"] or e["writes"] or e["calls"] or e["constructed"]):
            continue
        res[r] = {"what": e["what"], "reads": sorted(e["reads"]), "writes": sorted(e["writes"]),
                  "calls": {k: {"positional": sorted(v["positional"]), "keywords": sorted(v["keywords"])} for k, v in sorted(e["calls"].items())},
                  "constructed": e["constructed"]}
    return res


DEFINED = {"scripts/MPC_Wrapper.py": ["MPC_Wrapper", "Dummy"], "scripts/QP_WBC.py": ["wbc_controller"],
           "scripts/solo12InvKin.py": ["Solo12InvKin"]}


def walk_definitions(path, class_names):
    """What the reference's own classes DEFINE: per method the (min, max) number of positional arguments a caller may pass
    (self excluded) and the attributes assigned on self anywhere in the class -- so that a caller elsewhere in the reference
    that no longer matches its own class (scripts/test_mpc.py is such a file) can be told from a real requirement."""
    tree = ast.parse(open(path).read())
    out = {}
    for node in tree.body:
        if isinstance(node, ast.ClassDef) and node.name in class_names:
            methods, attrs = {}, set()
            for f in node.body:
                if isinstance(f, ast.FunctionDef):
                    a = f.args
                    npos = len(a.args) - 1
                    methods[f.name] = [npos - len(a.defaults), None if a.vararg else npos]
                    for n in ast.walk(f):
                        if isinstance(n, ast.Attribute) and isinstance(n.ctx, ast.Store) and isinstance(n.value, ast.Name) and n.value.id == "self":
                            attrs.add(n.attr)
            out[node.name] = {"methods": dict(sorted(methods.items())), "attributes": sorted(attrs)}
    return out


def walk_bindings(path):
    src = open(path).read()
    classes = {}
    # every visitor struct: template <typename X> struct XPythonVisitor ... { visit(...) { cl.def(...)... } expose() { bp::class_<X>("Name" ...
    for m in re.finditer(r"struct\s+(\w+)PythonVisitor\b(.*?)\n};", src, re.S):
        body = m.group(2)
        name_m = re.search(r'bp::class_<\s*\w+\s*>\(\s*"(\w+)"', body)
        if not name_m:
            continue
        ctors = []
        for c in re.finditer(r"bp::init<([^>]*)>", body):
            types = [t for t in (x.strip() for x in c.group(1).split(",")) if t]
            ctors.append(len(types))
        methods = {}
        defs = [(d.start(), d.group(1)) for d in re.finditer(r'\.def\(\s*"(\w+)"\s*,\s*&\w+::\w+', body)]
        for i, (pos, name) in enumerate(defs):
            seg = body[pos:(defs[i + 1][0] if i + 1 < len(defs) else len(body))]
            seg = seg.split(";")[0]  # the chain of .def() ends at the statement's semicolon
            a = re.search(r"bp::args\((.*?)\)", seg, re.S)
            methods[name] = len(re.findall(r'"[^"]*"', a.group(1))) if a else 0
        classes[name_m.group(1)] = {"constructors": sorted(set(ctors)), "methods": dict(sorted(methods.items()))}
    return classes


def main():
    surface = {"generated_by": "tests/golden/make_reference_surface.py", "reference": "paLeziart/quadruped-reactive-walking",
               "python": {}, "bindings": {"file": "python/gepadd.cpp", "classes": walk_bindings(os.path.join(REF, "python", "gepadd.cpp"))}}
    for rel, roots in PY_FILES.items():
        surface["python"][rel] = walk_python(os.path.join(REF, rel), roots)
    surface["definitions"] = {rel: walk_definitions(os.path.join(REF, rel), names) for rel, names in DEFINED.items()}
    json.dump(surface, open(OUT, "w"), indent=1, sort_keys=True)
    n = sum(len(v["reads"]) + len(v["writes"]) + len(v["calls"]) for f in surface["python"].values() for v in f.values())
    print("wrote %s: %d attribute / method entries from %d scripts, %d bound classes" % (OUT, n, len(PY_FILES), len(surface["bindings"]["classes"])))


if __name__ == "__main__":
    main()
