"""Empty stand-in so the reference's `import cv2` succeeds (visualisation only, never called)."""
