from . import cpn, cpn_original  # noqa: F401
