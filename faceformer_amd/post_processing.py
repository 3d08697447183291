"""Import-path parity with reference `faceformer/post_processing.py`; the implementations live in
`faceformer_amd.faces`."""
from .faces import (filter_faces_by_coedge, filter_faces_by_encloseness, is_face_enclosed,  # noqa: F401
                    map_coedge_into_edges)
