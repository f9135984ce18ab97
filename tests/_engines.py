"""Which engine configuration a GPU test runs with (see the spf_ctx fixture in conftest.py)."""
import pytest

both_engines = pytest.mark.parametrize("spf_ctx", ["default", "sweeps", "kfused", "lanevertex", "xcd"], indirect=True)
sweeps_engine = pytest.mark.parametrize("spf_ctx", ["sweeps"], indirect=True)
all_engines = pytest.mark.parametrize("spf_ctx", ["default", "sweeps", "kfused", "twophase", "widemask", "lanevertex", "xcd"], indirect=True)
hub_engines = pytest.mark.parametrize("spf_ctx", ["default", "hubsort", "patchfull"], indirect=True)
hubsort_engine = pytest.mark.parametrize("spf_ctx", ["hubsort"], indirect=True)
