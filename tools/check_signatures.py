#!/usr/bin/env python3
"""Signature check of the drop-in adaptors against the reference's REAL headers (build container only: /root/reference is not on the GPU box).

hyslam_amd/host/*.h compile here against host/cv_compat.h — this repository's own re-declaration of the hySLAM / OpenCV types the adaptors'
signatures mention — because OpenCV 3.4 and hySLAM cannot be built in this image.  Drift between cv_compat.h and the reference's headers would
go unnoticed by every other test, so this tool reads both as TEXT and fails when a member function that cv_compat.h declares (= what the adaptors
override or call) differs from the reference's declaration of the same class in name, arity, parameter types, const-ness or return type.

  * every class of cv_compat.h that mirrors a reference class is compared member function by member function: each compat declaration must
    have a reference overload with the same normalised signature (cv_compat is a SUBSET of the reference's interface);
  * for the classes the adaptors DERIVE from or REPLACE (FeatureExtractor, FeatureFactory, FeatureMatcher, Stereomatcher -> HipStereomatcher)
    the check is two-sided: every public member function of the reference must be present in the compat / adaptor class too;
  * `virtual` is compared as well: the only accepted differences are the ones INTEGRATION.md §3 documents as the header patch
    (FeatureMatcher's search entry points + destructor, FeatureFactory::getFeatureMatcher + destructor);
  * members that exist only in cv_compat.h (test set-up: public fields instead of Map / MapPointDB plumbing) are listed in COMPAT_ONLY with the reason.

It is a signature check: it pins no arithmetic, and nothing of the reference is copied — only declarations are parsed and compared.
usage: check_signatures.py [reference_root]      exit code 0 = all signatures agree; prints a report either way."""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
COMPAT = os.environ.get("HS_CHECK_COMPAT") or os.path.join(ROOT, "hyslam_amd", "host", "cv_compat.h")      # (the override: tests feed a mutated copy to see the check fire)

# (compat class, file that declares it here, reference header, reference class, two_sided)
PAIRS = [
    ("FeatureExtractor", COMPAT, "src/features/FeatureExtractor.h", "FeatureExtractor", True),
    # cv_compat.h declares these two classes twice: first AS THEY ARE in the reference (HYSLAM_AMD_COMPAT_UNPATCHED: the replacement translation unit
    # host/replace/FeatureMatcher.cc is compiled against that one; it must agree without any `virtual` allowance), then as they look after the patch
    ("FeatureFactory", COMPAT, "src/features/FeatureFactory.h", "FeatureFactory", True, 0),
    ("FeatureMatcher", COMPAT, "src/features/FeatureMatcher.h", "FeatureMatcher", True, 0),
    ("FeatureFactory", COMPAT, "src/features/FeatureFactory.h", "FeatureFactory", True, 1),
    ("FeatureMatcher", COMPAT, "src/features/FeatureMatcher.h", "FeatureMatcher", True, 1),
    ("HipStereomatcher", os.path.join(ROOT, "hyslam_amd", "host", "HipORBExtractor.h"), "src/features/Stereomatcher.h", "Stereomatcher", True),
    ("FeatureViews", COMPAT, "src/core/FeatureViews.h", "FeatureViews", False),
    ("FeatureDescriptor", COMPAT, "src/features/low_level/FeatureDescriptor.h", "FeatureDescriptor", False),
    ("DescriptorDistance", COMPAT, "src/features/low_level/DescriptorDistance.h", "DescriptorDistance", False),
    ("Frame", COMPAT, "src/core/Frame.h", "Frame", False),
    ("KeyFrame", COMPAT, "src/core/KeyFrame.h", "KeyFrame", False),
    ("MapPoint", COMPAT, "src/core/MapPoint.h", "MapPoint", False),
    ("LandMarkMatches", COMPAT, "src/core/LandMarkMatches.h", "LandMarkMatches", False),
    ("Camera", COMPAT, "src/core/Camera.h", "Camera", False),
]

# `virtual` differences that ARE the documented header patch: (class, member name)
VIRTUAL_PATCH = {("FeatureMatcher", n) for n in ("SearchByProjection", "SearchByBoW", "SearchByBoW2", "SearchForTriangulation", "SearchForInitialization",
                                                   "Fuse", "SearchBySim3", "~FeatureMatcher")} | {("FeatureFactory", "getFeatureMatcher"), ("FeatureFactory", "~FeatureFactory")}
# members of the compat classes with no counterpart in the reference header, and why that is fine
COMPAT_ONLY = {
    ("FeatureMatcher", "~FeatureMatcher"): "virtual destructor: part of the `virtual` patch (a class with virtual functions deleted through unique_ptr<FeatureMatcher>)",
    ("FeatureFactory", "~FeatureFactory"): "virtual destructor: part of the `virtual` patch",
    ("DescriptorDistance", "~DescriptorDistance"): "cv_compat only: the adaptors never delete through this base",
    ("HipStereomatcher", "HipStereomatcher"): "an ADDITIONAL constructor on an explicit C-ABI handle (the reference's (FeatureViews, Camera, FeatureMatcherSettings) one is there too and is checked)",
    ("HipStereomatcher", "setDefaultDevice"): "adaptor-only static: which GPU the per-thread handles are created on",
    ("Frame", "Frame"): "test set-up constructor (the reference builds Frames from image data)",
    ("KeyFrame", "KeyFrame"): "test set-up constructor",
    ("FeatureDescriptor", "FeatureDescriptor"): None,
}
# reference members the two-sided classes do not need (not part of the call surface the adaptors replace)
REF_ONLY_OK = {
    ("FeatureMatcher", "FeatureMatcher(float,bool)"): "(patched declaration only; the unpatched one has it) the (nnratio, checkOri) constructor leaves TH_LOW / TH_HIGH uninitialised and the factory never uses it (SURVEY quirk 6)",
    ("FeatureMatcher", "ComputeThreeMaxima"): "protected helper of the reference's own bodies",
    ("FeatureMatcher", "_SearchByProjection_"): "protected core of the reference's own bodies (replaced by the C ABI)",
    ("FeatureMatcher", "_SearchByBoW_"): "protected core of the reference's own bodies (replaced by the C ABI)",
}


def strip_comments(t):
    t = re.sub(r"/\*.*?\*/", " ", t, flags=re.S)
    t = re.sub(r"//[^\n]*", " ", t)
    t = re.sub(r"^\s*#[^\n]*", " ", t, flags=re.M)
    return t


def class_body(text, name, occurrence=0):
    """text between the braces of the `occurrence`-th definition of `class name` / `struct name` (definitions, not forward declarations)"""
    for k, m in enumerate(re.finditer(r"\b(class|struct)\s+" + re.escape(name) + r"\b([^;{]*)\{", text)):
        if k != occurrence:
            continue
        i = m.end()
        depth = 1
        j = i
        while j < len(text) and depth:
            depth += text[j] == "{"
            depth -= text[j] == "}"
            j += 1
        return text[i:j - 1], m.group(1)
    return None, None


def statements(body, kind):
    """top-level statements of a class body with their access level; inline function bodies and initialiser lists are dropped"""
    out, cur, depth_p, depth_a, access = [], "", 0, 0, ("private" if kind == "class" else "public")
    i = 0
    while i < len(body):
        c = body[i]
        if c == "{" and depth_p == 0:      # a function body (or a nested type / brace initialiser): skip it, the declaration ends here
            d = 1
            i += 1
            while i < len(body) and d:
                d += body[i] == "{"
                d -= body[i] == "}"
                i += 1
            if "(" in cur:
                out.append((access, cur.strip()))
                cur = ""
                while i < len(body) and body[i] in " \t\n;":
                    i += 1
            else:
                cur += " {} "
            continue
        if c == "(":
            depth_p += 1
        elif c == ")":
            depth_p -= 1
        elif c == "<":
            depth_a += 1
        elif c == ">":
            depth_a = max(0, depth_a - 1)
        if c == ":" and depth_p == 0 and re.fullmatch(r"\s*(public|private|protected)\s*", cur):
            access = cur.strip()
            cur = ""
        elif c == ";" and depth_p == 0:
            if cur.strip():
                out.append((access, cur.strip()))
            cur = ""
        else:
            cur += c
        i += 1
    return out


def split_top(s, sep=","):
    parts, cur, d = [], "", 0
    for c in s:
        if c in "<([":
            d += 1
        elif c in ">)]":
            d -= 1
        if c == sep and d == 0:
            parts.append(cur)
            cur = ""
        else:
            cur += c
    if cur.strip():
        parts.append(cur)
    return parts


def norm_type(t):
    t = re.sub(r"\s+", " ", t.strip())
    t = re.sub(r"\bstd::size_t\b", "size_t", t)
    t = re.sub(r"\b(inline|virtual|static|explicit|constexpr)\b", "", t)
    t = re.sub(r"\s*([&*<>,])\s*", r"\1", t).strip()
    t = re.sub(r"\s+", " ", t)
    return t


def norm_param(p):
    p = split_top(p, "=")[0].strip()                      # default value
    m = re.match(r"^(.*[\s&*>])([A-Za-z_]\w*)$", p)      # trailing parameter name
    if m and m.group(1).strip() and not re.fullmatch(r"(const|unsigned|signed|long|short)\s*", m.group(1)):
        p = m.group(1)
    return norm_type(p)


def parse_function(stmt, cls):
    """(name, return type, (param types), const, virtual) or None when the statement is not a member function declaration"""
    if "(" not in stmt or stmt.startswith(("using ", "typedef ", "friend ", "template")):
        return None
    stmt = re.sub(r"\s+", " ", stmt)
    if "operator()" in stmt:
        head, rest = stmt.split("operator()", 1)
        name = "operator()"
    else:
        m = re.search(r"(~?[A-Za-z_]\w*)\s*\(", stmt)
        if not m:
            return None
        name, head, rest = m.group(1), stmt[:m.start()], stmt[m.start() + len(m.group(1)):]
        if re.search(r"=\s*$", head) or re.search(r"[=]", head):      # `T x = f(...)`: a data member with an initialiser
            return None
    rest = rest.strip()
    if not rest.startswith("("):
        return None
    d, j = 0, 0
    for j, c in enumerate(rest):
        d += c == "("
        d -= c == ")"
        if d == 0:
            break
    params, tail = rest[1:j], rest[j + 1:]
    tail = tail.split(":")[0]                              # constructor initialiser list
    virtual = bool(re.search(r"\bvirtual\b", head)) or bool(re.search(r"\boverride\b", tail))
    const = bool(re.search(r"\bconst\b", tail))
    ret = norm_type(head)
    ptypes = tuple(norm_param(p) for p in split_top(params) if p.strip() and p.strip() != "void")
    if name in (cls, "~" + cls):
        ret = ""
    return name, ret, ptypes, const, virtual


def members(path, cls, public_only=False, occurrence=0):
    text = strip_comments(open(path, errors="replace").read())
    body, kind = class_body(text, cls, occurrence)
    if body is None:
        return None
    out = []
    for access, st in statements(body, kind):
        f = parse_function(st, cls)
        if f and (not public_only or access == "public"):
            out.append(f + (access,))
    return out


def sig(f):
    return "%s %s(%s)%s" % (f[1], f[0], ", ".join(f[2]), " const" if f[3] else "")


def main():
    if not os.path.isdir(REF):
        print("reference not present (%s): nothing to check" % REF)
        return 0
    problems, notes, checked = [], [], 0
    for pair in PAIRS:
        ccls, cfile, rhdr, rcls, two_sided = pair[:5]
        occ = pair[5] if len(pair) > 5 else 0
        unpatched = len(pair) > 5 and occ == 0
        rpath = os.path.join(REF, rhdr)
        cm, rm = members(cfile, ccls, occurrence=occ), members(rpath, rcls) if os.path.exists(rpath) else None
        if cm is None:
            problems.append("%s: class not found in %s" % (ccls, os.path.relpath(cfile, ROOT)))
            continue
        if rm is None:
            problems.append("%s: class %s not found in %s" % (ccls, rcls, rhdr))
            continue
        ren = lambda n: rcls if n == ccls else ("~" + rcls if n == "~" + ccls else n)       # HipStereomatcher <-> Stereomatcher
        for f in cm:
            name = ren(f[0])
            cands = [r for r in rm if r[0] == name]
            key = (ccls, f[0])
            if not cands:
                if key in COMPAT_ONLY and COMPAT_ONLY[key] and not (unpatched and ccls == "FeatureMatcher"):
                    notes.append("%s::%s — only here: %s" % (ccls, f[0], COMPAT_ONLY[key]))
                elif f[4 + 1] != "public" or re.match(r"^(m[A-Z]|n[A-Z]|size$)", f[0]):
                    pass
                else:
                    problems.append("%s::%s is not declared by %s (%s)" % (ccls, sig(f), rcls, rhdr))
                continue
            match = [r for r in cands if r[1:4] == f[1:4]]
            if not match:
                if key in COMPAT_ONLY and COMPAT_ONLY[key]:
                    notes.append("%s::%s — differs on purpose: %s" % (ccls, f[0], COMPAT_ONLY[key]))
                    continue
                problems.append("%s::%s\n      reference (%s): %s" % (ccls, sig(f), rhdr, " | ".join(sig(r) for r in cands)))
                continue
            checked += 1
            if match[0][4] != f[4]:
                if (ccls, f[0]) in VIRTUAL_PATCH and f[4] and not match[0][4] and not unpatched:
                    notes.append("%s::%s — `virtual` here, not in the reference: the documented patch (INTEGRATION.md §3)" % (ccls, f[0]))
                else:
                    problems.append("%s::%s: virtual = %s here, %s in the reference" % (ccls, sig(f), f[4], match[0][4]))
        if two_sided:
            for r in rm:
                if r[5] != "public" and (rcls, r[0]) not in {(c, n) for (c, n) in REF_ONLY_OK}:
                    continue
                name = ccls if r[0] == rcls else r[0]
                if any(f[0] == name and f[1:4] == r[1:4] for f in cm):
                    continue
                k1, k2 = (rcls, r[0]), (rcls, "%s(%s)" % (r[0], ",".join(r[2])))
                why = REF_ONLY_OK.get(k2) or REF_ONLY_OK.get(k1)
                if why:
                    notes.append("%s::%s — reference only: %s" % (rcls, sig(r), why))
                elif r[5] == "public":
                    problems.append("%s::%s (%s) has no counterpart in %s" % (rcls, sig(r), rhdr, ccls))
    print("check_signatures: %d member functions of %d classes agree with %s" % (checked, len(PAIRS), REF))
    for n in notes:
        print("  note: " + n)
    for p in problems:
        print("  MISMATCH: " + p)
    return 1 if problems else 0


if __name__ == "__main__":
    sys.exit(main())
