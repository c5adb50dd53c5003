"""Tiny text-level reader of a Rust source file's API surface: `use` imports and `pub fn` signatures (no parsing beyond
balanced parentheses).  Shared by tests/golden/make_api_signatures.py (reference side, run once here) and
tests/test_abi_and_host.py (rust/src/*.rs side)."""
import re


def strip_comments(t):
    return re.sub(r"//[^\n]*", "", re.sub(r"/\*.*?\*/", "", t, flags=re.S))


def imports(text):
    """`use a::b::{C, D};` -> {C: a::b, D: a::b}"""
    out = {}
    for m in re.finditer(r"^\s*use\s+([^;]+);", text, re.M):
        path = re.sub(r"\s+", "", m.group(1))
        g = re.match(r"(.*)::\{(.*)\}$", path)
        if g:
            for name in g.group(2).split(","):
                if name:
                    out[name.split("::")[-1]] = g.group(1) + ("::" + "::".join(name.split("::")[:-1]) if "::" in name else "")
        else:
            parts = path.split("::")
            out[parts[-1]] = "::".join(parts[:-1])
    return out


def canon(sig):
    """whitespace-insensitive form of a signature: single spaces between words, none around punctuation, no trailing comma"""
    s = re.sub(r"\s+", " ", sig).strip()
    s = re.sub(r"\s*([(),:<>\[\]&])\s*", r"\1", s)
    return s.replace(",)", ")")


def signatures(text):
    """{impl type or '': {fn name: flattened signature up to the body}}"""
    out = {}
    impls = [(m.start(), m.group(1)) for m in re.finditer(r"^impl(?:<[^>]*>)?\s+(?:[\w:<>, ]+\s+for\s+)?(\w+)", text, re.M)]
    tests_at = text.find("#[cfg(test)]")
    for m in re.finditer(r"pub fn\s+(\w+)\s*(?:<[^>]*>)?\s*\(", text):
        if 0 <= tests_at < m.start():
            break
        depth, i = 0, m.end() - 1
        while True:
            depth += text[i] == "("
            depth -= text[i] == ")"
            i += 1
            if depth == 0:
                break
        j = text.index("{", i)
        sig = canon(text[m.start():j])
        owner = ""
        for pos, name in impls:
            if pos < m.start():
                owner = name
        # a free function after the last impl block: owner only if it is indented (inside the impl)
        line_start = text.rfind("\n", 0, m.start()) + 1
        if text[line_start:m.start()].strip() == "" and m.start() == line_start:
            owner = ""
        out.setdefault(owner, {})[m.group(1)] = sig
    return out


def mod_decls(text):
    """names declared by `mod x;` / `pub mod x;` (declarations with a body `mod x { .. }` are inline modules: not files)"""
    return re.findall(r"^\s*(?:pub(?:\([^)]*\))?\s+)?mod\s+(\w+)\s*;", text, re.M)


def pub_items(text):
    """names of the top-level pub items of a module's text: struct / enum / fn / const / static / trait / type / mod, and what its
    `pub use` lines re-export.  Top level = the keyword `pub` starts its line (items inside impl / mod blocks are indented)."""
    names = set(re.findall(r"^pub\s+(?:unsafe\s+)?(?:struct|enum|fn|const|static|trait|type|mod)\s+(?:mut\s+)?(\w+)", text, re.M))
    for m in re.finditer(r"^pub\s+use\s+([^;]+);", text, re.M):
        path = re.sub(r"\s+", "", m.group(1))
        g = re.match(r"(.*)::\{(.*)\}$", path)
        for n in (g.group(2).split(",") if g else [path]):
            if n:
                names.add(n.split("::")[-1])
    return sorted(names)
