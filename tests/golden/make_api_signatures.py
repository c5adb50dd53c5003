"""Extract the reference crate's API surface for the hot path as DATA (tests/golden/reference_api_signatures.json): per file
the `use` imports (name -> path) and the flattened signature of every `pub fn`, keyed by the impl type it belongs to.
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


def main():
    api = {}
    for short, rel in FILES.items():
        t = strip_comments(open(os.path.join(REF, rel)).read())
        t = t[:t.find("#[cfg(test)]")] if "#[cfg(test)]" in t else t          # the crate's API, not its test modules
        api[short] = {"source": "src/" + rel, "imports": imports(t), "pub_fn": signatures(t)}
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_api_signatures.json")
    json.dump(api, open(out, "w"), indent=1, sort_keys=True)
    print(out, {k: {o: sorted(v) for o, v in a["pub_fn"].items()} for k, a in api.items()})


if __name__ == "__main__":
    main()
