"""Drivers around the hot path: the hierarchical keyframe -> clip alignment and its output writers."""
