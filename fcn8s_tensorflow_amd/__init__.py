"""MI355X-native FCN-8s hot path behind the reference's FCN8s / BatchGenerator API."""
__version__ = "0.1.0"
