"""Algorithm constants (reference: src/constants.py:1-5)."""
MAX_ITER = 30
LAMBDA_0 = 80.0
LAMBDA_N = 5.0
LAMBDA_RATIO = 0.9

ZOOM_SIGMA_ZERO = 0.6  # unused by the reference's pyramid (it calls skimage.rescale); kept for API parity
