"""OBJ ingest ORACLE (test infrastructure, NOT product code): pure-Python restatement
of what obj::load (engine/src/obj.rs:44-151) obtains from the `tobj` crate.

tobj is a third-party dependency (engine/Cargo.toml:8, version "*", 3.x API as used
at obj.rs:45-52) whose source is NOT under /root/reference, and the reference's only
test of this path is `load(..).is_some()` (obj.rs:229-233): parity for the OBJ path
is therefore UNPINNED.  This file restates tobj 3.x's published behaviour
(load_obj_buf) as far as obj.rs consumes it:

  * `v x y z`     -> three f32 (Rust `str::parse::<f32>`, correctly rounded), later
                     widened with `as f64` (obj.rs:103-105);
  * `f a b c ...` -> indices; i > 0 is 1-based, i < 0 counts back from the vertices
                     read so far; `a/b/c` forms keep the position index;
  * `o` / `g`     -> closes the current model if it has faces, then renames;
  * `usemtl m`    -> closes the current model if the material id changes and the
                     model has faces;
  * `mtllib f`    -> every library must load (obj.rs:64 unwraps with "WOOPS");
  * end of file   -> the current model is emitted unconditionally;
  * triangulate   -> fan (v0, v[i], v[i+1]); 1- and 2-vertex faces dropped.

Written independently of csrc/rm_scene.cpp (which is the product's loader) so that
the two can be compared in tests/test_obj_ingest.py.
"""
import os
import struct
from fractions import Fraction


def parse_f32(token):
    """Decimal string -> nearest binary32 (ties to even), returned as a Python float
    holding exactly that value.  Exact (no double rounding): candidates around the
    double approximation are compared as rationals."""
    approx = float(token)
    if approx != approx or approx in (float("inf"), float("-inf")):
        return approx
    exact = Fraction(token) if not any(c in token for c in "eE") else Fraction(token.replace("E", "e"))
    f = struct.unpack("<f", struct.pack("<f", approx))[0]
    bits = struct.unpack("<i", struct.pack("<f", f))[0]
    best = None
    for delta in (-1, 0, 1):
        b = bits + delta
        try:
            cand = struct.unpack("<f", struct.pack("<i", b))[0]
        except struct.error:
            continue
        if cand != cand or cand in (float("inf"), float("-inf")):
            continue
        if (cand < 0) != (f < 0) and cand != 0 and f != 0:
            continue
        err = abs(Fraction(cand) - exact)
        key = (err, b & 1)            # nearest; on a tie the even mantissa
        if best is None or key < best[0]:
            best = (key, cand)
    return best[1]


def _load_mtl(path, mat_map, counter):
    with open(path, "r") as f:      # raises -> the reference panics "WOOPS" (obj.rs:64)
        for line in f:
            words = line.split()
            if not words or words[0] == "#":
                continue
            if words[0] == "newmtl":
                name = line.strip()[len("newmtl"):].strip()
                if not name:
                    raise ValueError("newmtl without a name")
                mat_map[name] = counter[0]
                counter[0] += 1


def load_models(path):
    """-> list of (name, [triangle]) with triangle = 9 floats (f32 values as doubles)."""
    base = os.path.dirname(path)
    positions = []            # flat f32 list
    faces = []                # faces of the model being read: lists of resolved indices
    models = []
    name = "unnamed_object"
    mat_map, counter = {}, [0]
    mat_id = None

    def export():
        tris = []
        for face in faces:
            if len(face) < 3:
                continue      # ignore_points / ignore_lines
            a = face[0]
            b = face[1]
            for c in face[2:]:
                tri = []
                for idx in (a, b, c):
                    tri.extend(positions[3 * idx:3 * idx + 3])
                tris.append(tri)
                b = c
        models.append((name, tris))
        del faces[:]

    with open(path, "r") as f:
        for line in f:
            words = line.split()
            if not words or words[0] == "#":
                continue
            kw = words[0]
            if kw == "v":
                vals = [parse_f32(t) for t in words[1:4]]
                if len(vals) != 3:
                    raise ValueError("PositionParseError")
                positions.extend(vals)
            elif kw in ("f", "l"):
                n_pos = len(positions) // 3
                face = []
                for tok in words[1:]:
                    i = int(tok.split("/")[0])
                    i = n_pos + i if i < 0 else i - 1
                    if not 0 <= i < n_pos:
                        raise ValueError("FaceParseError")
                    face.append(i)
                if not face:
                    raise ValueError("FaceParseError")
                faces.append(face)
            elif kw in ("o", "g"):
                if faces:
                    export()
                name = line.strip()[1:].strip() or "unnamed_object"
            elif kw == "mtllib":
                _load_mtl(os.path.join(base, words[1]), mat_map, counter)
            elif kw == "usemtl":
                mat_name = line.strip()[len("usemtl"):].strip()
                if not mat_name:
                    raise ValueError("MaterialParseError")
                new_mat = mat_map.get(mat_name)
                if mat_id != new_mat and faces:
                    export()
                mat_id = new_mat
    export()
    return models
