"""The two environment variables of the package (csrc/route.h): SEERHIP_ROUTE="key=value,..." forces a route a run would otherwise choose
from its data (a test hook; the library refuses unknown keys), SEERHIP_DEBUG="item,item" turns diagnostics on stderr on."""
import os


def route(key, default=None):
    for item in os.environ.get("SEERHIP_ROUTE", "").split(","):
        k, eq, v = item.partition("=")
        if eq and k == key:
            return v
    return default


def debug(item):
    return item in os.environ.get("SEERHIP_DEBUG", "").split(",")


def with_route(env_value, **kv):
    """env_value (the current SEERHIP_ROUTE string or None) with the given keys set (value None: removed) -> the new string."""
    items = [it for it in (env_value or "").split(",") if it and it.partition("=")[0] not in kv]
    items += ["%s=%s" % (k, v) for k, v in kv.items() if v is not None]
    return ",".join(items)
