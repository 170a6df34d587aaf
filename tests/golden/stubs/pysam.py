"""stub: the pinned lines never call pysam"""
