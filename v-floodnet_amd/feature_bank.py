class FeatureBank:  # placeholder, replaced below
    pass
