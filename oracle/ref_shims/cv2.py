"""Stand-in for the un-installed `cv2`: Canny only (oracle/canny.c)."""
from oracle.canny import Canny  # noqa: F401
