"""The chunk record exchanged with the indexers (schema of the reference's
``Chunk.to_dict``, ``src/preprocessing/chunker.py:9-26``)."""
from dataclasses import dataclass, field, asdict
from typing import Dict


@dataclass
class Chunk:
    chunk_id: str
    doc_id: str
    source: str
    chunk_type: str
    text: str
    metadata: Dict = field(default_factory=dict)

    def to_dict(self):
        d = asdict(self)
        d['metadata'] = self.metadata        # keep the caller's object, like the reference
        return d
