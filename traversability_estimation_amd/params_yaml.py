"""Reads the reference's parameter files UNCHANGED into a te_params block (+ the run flags the chain's tail implies).

What the ROS node does with these files (reference, file:line):
  * `robot_filter_parameter.yaml` -- the list `traversability_map_filters` is what `filters::FilterChain::configure`
    walks (`TE/src/TraversabilityMap.cpp:129-131`); each entry's `params` are read by the plugin's `configure()`
    (`TEF/src/SlopeFilter.cpp:34-56`, `StepFilter.cpp:38-99`, `RoughnessFilter.cpp:36-70`).
  * `robot_footprint_parameter.yaml` -- `footprint/...` (`TE/src/TraversabilityMap.cpp:92-126`).
  * `robot.yaml` -- `max_gap_width` (`TraversabilityMap.cpp:117`).

The chain this library runs is the fixed sequence normals -> slope -> step -> roughness -> weighted sum -> deletion; a file
that asks for anything else (another order, another filter type, an expression that is not a weighted sum of the three
scores, an input layer other than `elevation`) is refused with a message rather than half-applied.  Host-side plumbing
for bench.py and the tests: the plugins themselves get their parameters from `FilterBase::getParam`, as in the reference.
"""
import re

import numpy as np

CHAIN_KEY = "traversability_map_filters"
_NUM = r"[-+]?(?:\d+\.?\d*(?:[eE][-+]?\d+)?|\.\d+(?:[eE][-+]?\d+)?)"
_SCORES = {"traversability_slope": "w_slope", "traversability_step": "w_step", "traversability_roughness": "w_rough"}


class ParamsYamlError(ValueError):
    pass


def _load(path_or_text):
    import yaml
    if "\n" not in str(path_or_text):
        with open(path_or_text) as f:
            return yaml.safe_load(f)
    return yaml.safe_load(path_or_text)


def _scalar_f32(text):
    """`1.0 / 3.0`, `0.25`, `(1.0/3.0)`: evaluated in float32 like EigenLab does on MatrixXf scalars."""
    t = text.strip()
    while t.startswith("(") and t.endswith(")"):
        t = t[1:-1].strip()
    m = re.fullmatch(rf"({_NUM})\s*/\s*({_NUM})", t)
    if m:
        return np.float32(np.float32(float(m.group(1))) / np.float32(float(m.group(2))))
    if re.fullmatch(_NUM, t):
        return np.float32(float(t))
    raise ParamsYamlError(f"weighted sum: cannot read the scalar '{text}'")


def parse_weighted_sum(expression, names=_SCORES):
    """`scale * (a + b + c)` with optional per-layer factors `w * layer`; returns {"w_scale", "w_slope", "w_step", "w_rough"}
    as float32.  The shipped expression (`robot_filter_parameter.yaml:33`) gives (1.0f/3.0f, 1, 1, 1)."""
    e = expression.strip()
    m = re.fullmatch(r"(.+?)\*\s*\((.+)\)", e)
    scale, body = (np.float32(1.0), e) if not m else (_scalar_f32(m.group(1)), m.group(2))
    if not m and e.startswith("(") and e.endswith(")"):
        body = e[1:-1]
    out = {"w_scale": scale}
    order = []
    for term in body.split("+"):
        t = term.strip()
        mm = re.fullmatch(rf"(?:(.+?)\*\s*)?([A-Za-z_][A-Za-z_0-9]*)", t)
        if not mm or mm.group(2) not in names:
            raise ParamsYamlError(f"weighted sum: '{t}' is not a (weighted) score layer of the chain")
        key = names[mm.group(2)]
        if key in out:
            raise ParamsYamlError(f"weighted sum: layer '{mm.group(2)}' appears twice")
        out[key] = np.float32(1.0) if mm.group(1) is None else _scalar_f32(mm.group(1))
        order.append(key)
    if order != ["w_slope", "w_step", "w_rough"]:
        # float32 addition is not associative: the kernel evaluates ((slope + step) + roughness), left to right like EigenLab
        raise ParamsYamlError("weighted sum: the three score layers must appear once each, in the order slope + step + roughness")
    return out


def _axis(v):
    try:
        return {"x": 0, "y": 1, "z": 2}[str(v).strip().lower()]
    except KeyError:
        raise ParamsYamlError(f"normal_vector_positive_axis must be x, y or z (got '{v}')")


def filter_chain_fields(doc):
    """te_params fields (plain dict) + {"keep_normals": bool} from the `traversability_map_filters` list."""
    if not isinstance(doc, dict) or CHAIN_KEY not in doc:
        raise ParamsYamlError(f"no '{CHAIN_KEY}' list in the filter parameter file")
    chain = doc[CHAIN_KEY]
    want = ["gridMapFilters/NormalVectorsFilter", "traversabilityFilters/SlopeFilter", "traversabilityFilters/StepFilter",
            "traversabilityFilters/RoughnessFilter", "gridMapFilters/MathExpressionFilter"]
    types = [str(f.get("type")) for f in chain]
    if types[:5] != want or types[5:] not in ([], ["gridMapFilters/DeletionFilter"]):
        raise ParamsYamlError("the filter chain must be NormalVectorsFilter, SlopeFilter, StepFilter, RoughnessFilter, MathExpressionFilter"
                              f" [, DeletionFilter] in this order (got {types})")
    prm = [f.get("params") or {} for f in chain]
    nrm, slope, step, rough, comb = prm[:5]

    def need(d, key, who):
        if key not in d:
            raise ParamsYamlError(f"{who} did not find param {key}")  # the reference's ROS_ERROR wording
        return d[key]

    if str(nrm.get("input_layer", "elevation")) != "elevation":
        raise ParamsYamlError("NormalVectorsFilter: input_layer must be 'elevation'")
    prefix = str(nrm.get("output_layers_prefix", "surface_normal_"))
    if prefix != "surface_normal_":
        raise ParamsYamlError("NormalVectorsFilter: output_layers_prefix must be 'surface_normal_' (SlopeFilter.cpp:71 hard-codes it)")
    if str(nrm.get("algorithm", "area")) != "area":
        raise ParamsYamlError("NormalVectorsFilter: only the area method is built")
    out = {"normals_radius": float(need(nrm, "radius", "Normal vectors filter")), "normals_axis": _axis(nrm.get("normal_vector_positive_axis", "z")),
           "slope_critical": float(need(slope, "critical_value", "SlopeFilter")),
           "step_critical": float(need(step, "critical_value", "Step filter")),
           "step_radius1": float(need(step, "first_window_radius", "Step filter")),
           "step_radius2": float(need(step, "second_window_radius", "Step filter")),
           "step_ncrit": int(need(step, "critical_cell_number", "Step filter")),
           "rough_critical": float(need(rough, "critical_value", "Roughness filter")),
           "rough_radius": float(need(rough, "estimation_radius", "Roughness filter"))}
    for d, dflt in ((slope, "traversability_slope"), (step, "traversability_step"), (rough, "traversability_roughness")):
        if str(d.get("map_type", dflt)) != dflt:
            raise ParamsYamlError(f"map_type must stay '{dflt}': the footprint checks read the layers by these names (TraversabilityMap.cpp:52-56)")
    if str(comb.get("output_layer", "traversability")) != "traversability":
        raise ParamsYamlError("MathExpressionFilter: output_layer must be 'traversability'")
    out.update({k: float(v) for k, v in parse_weighted_sum(str(need(comb, "expression", "MathExpressionFilter"))).items()})
    deleted = set()
    if len(chain) == 6:
        deleted = {str(s) for s in (prm[5].get("layers") or [])}
        extra = deleted - {"surface_normal_x", "surface_normal_y", "surface_normal_z"}
        if extra:
            raise ParamsYamlError(f"DeletionFilter: only the normal layers can be dropped (got {sorted(extra)})")
    out["keep_normals"] = len(deleted) != 3
    # criticalStepHeight_ of the footprint checks is the step filter's critical value (TraversabilityMap.cpp:104-111)
    out["fp_critical_step"] = out["step_critical"]
    return out


def footprint_fields(doc):
    fp = (doc or {}).get("footprint")
    if not isinstance(fp, dict):
        raise ParamsYamlError("no 'footprint' block in the footprint parameter file")
    out = {}
    if "circular_footprint_radius_inscribed" in fp:
        out["fp_radius"] = float(fp["circular_footprint_radius_inscribed"])  # radiusMin of traversabilityFootprint (:348)
    if "circular_footprint_offset" in fp:
        out["fp_offset"] = float(fp["circular_footprint_offset"])
    if "traversability_default" in fp:
        out["fp_default"] = float(fp["traversability_default"])
    if "verify_roughness_footprint" in fp:
        out["fp_check_roughness"] = 1 if fp["verify_roughness_footprint"] else 0
    return out


def robot_fields(doc):
    out = {}
    if isinstance(doc, dict) and "max_gap_width" in doc:
        out["fp_max_gap"] = float(doc["max_gap_width"])
    return out


def fields_from_yaml(filter_yaml, footprint_yaml=None, robot_yaml=None):
    f = filter_chain_fields(_load(filter_yaml))
    if footprint_yaml is not None:
        f.update(footprint_fields(_load(footprint_yaml)))
    if robot_yaml is not None:
        f.update(robot_fields(_load(robot_yaml)))
    return f


def params_from_yaml(capi, filter_yaml, footprint_yaml=None, robot_yaml=None):
    """(te_params validated by the library, run flags): the reference's files in, what te_set_params / te_run_chain take out."""
    f = fields_from_yaml(filter_yaml, footprint_yaml, robot_yaml)
    keep = f.pop("keep_normals")
    p = capi.default_params(**f)
    capi.validate_params(p)
    return p, (capi.RUN_KEEP_NORMALS if keep else 0)
