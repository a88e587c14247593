from . import correlation, flow  # noqa: F401
