"""Inert stub: enet is out of scope."""
def glmnet_python(*a, **k):
    raise RuntimeError('glmnet not available')
