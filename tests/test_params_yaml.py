"""The reference's parameter files, read unchanged: they must reproduce te_params_default
field for field, and files the fixed chain cannot honour must be refused, not half-applied."""
import os

import pytest

from traversability_estimation_amd import params_yaml as Y

REF_CFG = "/root/reference/traversability_estimation/config"

# the shipped chain, restated as data (robot_filter_parameter.yaml:1-37 holds the same keys and values; the test below
# reads the reference's own file wherever /root/reference exists)
SHIPPED = """
traversability_map_filters:
  - {name: surfaceNormalsFilter, type: gridMapFilters/NormalVectorsFilter,
     params: {input_layer: elevation, output_layers_prefix: surface_normal_, radius: 0.05, normal_vector_positive_axis: z}}
  - {name: slopeFilter, type: traversabilityFilters/SlopeFilter, params: {map_type: traversability_slope, critical_value: 1.0}}
  - {name: stepFilter, type: traversabilityFilters/StepFilter,
     params: {map_type: traversability_step, critical_value: 0.12, first_window_radius: 0.04, second_window_radius: 0.04, critical_cell_number: 4}}
  - {name: roughnessFilter, type: traversabilityFilters/RoughnessFilter,
     params: {map_type: traversability_roughness, critical_value: 0.05, estimation_radius: 0.05}}
  - {name: weightedSumFilter, type: gridMapFilters/MathExpressionFilter,
     params: {output_layer: traversability, expression: (1.0 / 3.0) * (traversability_slope + traversability_step + traversability_roughness)}}
  - {name: deletionFilter, type: gridMapFilters/DeletionFilter, params: {layers: [surface_normal_x, surface_normal_y, surface_normal_z]}}
"""
FOOTPRINT = """
footprint:
  circular_footprint_radius_inscribed: 0.30
  circular_footprint_offset: 0.15
  traversability_default: 0.3
  verify_roughness_footprint: false
"""


@pytest.fixture(scope="module")
def capi():
    from traversability_estimation_amd import capi
    capi.load()
    return capi


def _same(capi, p, q):
    return capi.params_to_bytes(p) == capi.params_to_bytes(q)


def test_shipped_values_reproduce_the_library_defaults(capi):
    p, flags = Y.params_from_yaml(capi, SHIPPED, FOOTPRINT, "max_gap_width: 0.3\n")
    assert _same(capi, p, capi.default_params()) and flags == 0


@pytest.mark.skipif(not os.path.isdir(REF_CFG), reason="the reference tree is not on this box")
def test_the_references_own_files_reproduce_the_library_defaults(capi):
    p, flags = Y.params_from_yaml(capi, os.path.join(REF_CFG, "robot_filter_parameter.yaml"),
                                  os.path.join(REF_CFG, "robot_footprint_parameter.yaml"), os.path.join(REF_CFG, "robot.yaml"))
    d = capi.default_params()
    for f, _ in d._fields_:
        assert getattr(p, f) == getattr(d, f), f
    assert _same(capi, p, d) and flags == 0


def test_weighted_sum_forms():
    import numpy as np
    w = Y.parse_weighted_sum("(1.0 / 3.0) * (traversability_slope + traversability_step + traversability_roughness)")
    assert w["w_scale"] == np.float32(1.0) / np.float32(3.0) and w["w_slope"] == w["w_step"] == w["w_rough"] == 1.0
    w = Y.parse_weighted_sum("0.5 * (2 * traversability_slope + traversability_step + 0.25*traversability_roughness)")
    assert (w["w_scale"], w["w_slope"], w["w_step"], w["w_rough"]) == (0.5, 2.0, 1.0, 0.25)
    w = Y.parse_weighted_sum("traversability_slope + traversability_step + traversability_roughness")
    assert w["w_scale"] == 1.0
    for bad in ("traversability_slope * traversability_step", "(1/3) * (traversability_step + traversability_slope + traversability_roughness)",
                "0.5 * (traversability_slope + traversability_step)", "sqrt(traversability_slope) + traversability_step + traversability_roughness"):
        with pytest.raises(Y.ParamsYamlError):
            Y.parse_weighted_sum(bad)


def test_files_the_fixed_chain_cannot_honour_are_refused(capi):
    import yaml
    doc = yaml.safe_load(SHIPPED)
    swapped = {"traversability_map_filters": [doc["traversability_map_filters"][k] for k in (0, 2, 1, 3, 4, 5)]}
    with pytest.raises(Y.ParamsYamlError, match="in this order"):
        Y.filter_chain_fields(swapped)
    missing = yaml.safe_load(SHIPPED)
    del missing["traversability_map_filters"][2]["params"]["critical_cell_number"]
    with pytest.raises(Y.ParamsYamlError, match="did not find param critical_cell_number"):
        Y.filter_chain_fields(missing)
    kept = yaml.safe_load(SHIPPED)
    kept["traversability_map_filters"].pop()  # no DeletionFilter: the normals stay in the map
    assert Y.filter_chain_fields(kept)["keep_normals"] is True
    # range checks are the library's (the reference's configure() messages)
    bad = yaml.safe_load(SHIPPED)
    bad["traversability_map_filters"][1]["params"]["critical_value"] = 2.0
    with pytest.raises(capi.TeError, match="Critical slope"):
        Y.params_from_yaml(capi, yaml.safe_dump(bad) + "\n")
