"""Inert stub: enet is out of scope."""
def cvglmnetCoef(*a, **k):
    raise RuntimeError('glmnet not available')
