"""Extract the reference crate's API surface for the hot path as DATA (tests/golden/reference_api_signatures.json): per file
the `use` imports (name -> path) and the flattened signature of every `pub fn`, keyed by the impl type it belongs to; and
(round 4) the crate's MODULE TREE as reachable from src/lib.rs — module path -> names of its top-level pub items — plus the
imports of src/main.rs, so that the test can check that every `use crate::...` path of rust/src/**/*.rs resolves.
Run here (the reference is read as text; nothing of it is executed); tests/test_abi_and_host.py checks rust/src/*.rs against
the committed JSON, so the GPU box needs no reference checkout."""
import json
import os
import re
import sys

REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference/src"
FILES = {"do_acquisition.rs": "acquisition/do_acquisition.rs", "doppler_shift.rs": "acquisition/doppler_shift.rs",
         "do_tracking.rs": "tracking/do_tracking.rs", "fft.rs": "fft.rs"}


sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rs_api import imports, signatures, strip_comments  # noqa: E402


def module_tree(src_root):
    """{module path: sorted names of its top-level `pub` items (struct / enum / fn / const / static / trait / type, and the names a
    `pub use` re-exports)} for every module reachable from lib.rs through `mod x;` declarations (x.rs or x/mod.rs)."""
    import re as _re
    from rs_api import pub_items, mod_decls
    tree = {}

    def visit(path, file, dirpath):
        if not os.path.exists(file):
            return
        t = strip_comments(open(file).read())
        tree[path] = pub_items(t)
        for name in mod_decls(t):
            cand = [(os.path.join(dirpath, name + ".rs"), os.path.join(dirpath, name)),
                    (os.path.join(dirpath, name, "mod.rs"), os.path.join(dirpath, name))]
            for f, d in cand:
                if os.path.exists(f):
                    visit(path + "::" + name, f, d)
                    break
            else:
                tree[path + "::" + name] = None          # declared, file absent from the checkout
    visit("crate", os.path.join(src_root, "lib.rs"), src_root)
    return tree


def main():
    api = {}
    for short, rel in FILES.items():
        t = strip_comments(open(os.path.join(REF, rel)).read())
        t = t[:t.find("#[cfg(test)]")] if "#[cfg(test)]" in t else t          # the crate's API, not its test modules
        api[short] = {"source": "src/" + rel, "imports": imports(t), "pub_fn": signatures(t)}
    from rs_api import mod_decls
    api["_module_tree"] = module_tree(REF)
    api["_lib_rs_mods"] = mod_decls(strip_comments(open(os.path.join(REF, "lib.rs")).read()))
    api["_main_rs_imports"] = imports(strip_comments(open(os.path.join(REF, "main.rs")).read()))
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_api_signatures.json")
    json.dump(api, open(out, "w"), indent=1, sort_keys=True)
    print(out, {k: {o: sorted(v) for o, v in a["pub_fn"].items()} for k, a in api.items() if not k.startswith("_")})
    print({k: (len(v) if v is not None else None) for k, v in api["_module_tree"].items()})


if __name__ == "__main__":
    main()
