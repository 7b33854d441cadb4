"""Wire formats either side of the hot path, compatible with the reference's
(src/json-yaml.jl): `toDict` / `exportJSON` for shapes and `readconfig` for YAML configs."""
import json

import numpy as np

from .api import (DEFAULT_PARAMETERS, DEFAULT_SHAPE_DICT, ExtractedShape, FittedCone, FittedCylinder, FittedPlane,
                  FittedShape, FittedSphere, ransacparameters, strt)

# field order = fieldnames(typeof(s)) of the reference structs (shapes/*.jl)
_FIELDS = {FittedPlane: ("point", "normal"), FittedSphere: ("center", "radius", "outwards"),
           FittedCylinder: ("axis", "center", "radius", "outwards"), FittedCone: ("apex", "axis", "opang", "outwards")}


def toDict(s):
    """toDict(s) (json-yaml.jl:9-30): {"type": strt(s), <field>: value ...}; a list of shapes /
    ExtractedShapes becomes {"primitives": [...]}."""
    if isinstance(s, (list, tuple)):
        return {"primitives": [toDict(x) for x in s]}
    if isinstance(s, ExtractedShape):
        return toDict(s.shape)
    if not isinstance(s, FittedShape):
        raise TypeError("toDict expects a FittedShape, an ExtractedShape or a list of them")
    d = {"type": strt(s)}
    for f in _FIELDS[type(s)]:
        v = getattr(s, f)
        d[f] = [float(x) for x in v] if isinstance(v, np.ndarray) else v
    return d


def exportJSON(io, s, indent=None):
    """exportJSON(io, s[, indent]) (json-yaml.jl:43-59)."""
    if indent is None:
        io.write(json.dumps(toDict(s), separators=(",", ":")))
    else:
        io.write(json.dumps(toDict(s), indent=indent) + "\n")


def _dict2nt(v):
    """dict2nt (json-yaml.jl:98-109): a YAML list of single-key maps merged into one dict."""
    out = {}
    for d in v:
        out.update(d)
    return out


def readconfig(fname, toextend=None, shapedict=None):
    """readconfig(fname; toextend=DEFAULT_PARAMETERS, shapedict=DEFAULT_SHAPE_DICT) (json-yaml.jl:81-96)."""
    import yaml
    toextend = DEFAULT_PARAMETERS if toextend is None else toextend
    shapedict = DEFAULT_SHAPE_DICT if shapedict is None else shapedict
    with open(fname, "r", encoding="utf-8") as f:
        fdict = yaml.safe_load(f)
    for k, v in fdict.items():
        nt = _dict2nt(v)
        if "shape_types" in nt:
            nt = dict(nt)
            nt["shape_types"] = [shapedict[name] for name in nt["shape_types"]]
        toextend = ransacparameters(toextend, **{k: nt})
    return toextend
