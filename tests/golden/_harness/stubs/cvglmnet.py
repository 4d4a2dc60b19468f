"""Inert stub: enet is out of scope."""
def cvglmnet(*a, **k):
    raise RuntimeError('glmnet not available')
