"""Inert stub: enet is out of scope."""
def cvglmnetPredict(*a, **k):
    raise RuntimeError('glmnet not available')
