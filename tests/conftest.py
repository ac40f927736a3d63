import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    import refbind
    refbind.register_host_tree_builder()


GOLDEN = os.path.join(ROOT, "tests", "golden")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session")
def oracle():
    import refbind
    if not os.path.exists(refbind.ORACLE_LIB):
        pytest.fail("oracle/libpt_oracle.so is not built (run __graft_entry__.build())")
    return refbind.Oracle()


@pytest.fixture(scope="session")
def reference():
    """The compiled reference; present wherever oracle/_ref was built (it travels with the tree)."""
    import refbind
    if not os.path.exists(refbind.REF_LIB):
        pytest.skip("oracle/_ref/libpt_ref.so not present")
    return refbind.Reference()
