"""Amino-acid vocabulary, mirroring the reference's `protein/Sequence.py` API.

Reference: /root/reference/protein_transformer/protein/Sequence.py:1-91.  Ids: the 20
standard residues in alphabetical 1-letter order -> 0..19, '_' pad -> 20, '?' unknown -> 21;
with `add_sos_eos` '<' -> 22 and '>' -> 23, otherwise both map to the unknown id
(so `sos_id == eos_id == 21` for the default instance).
"""

ONE_TO_THREE_LETTER_MAP = {"A": "ALA", "C": "CYS", "D": "ASP", "E": "GLU", "F": "PHE", "G": "GLY", "H": "HIS",
                           "I": "ILE", "K": "LYS", "L": "LEU", "M": "MET", "N": "ASN", "P": "PRO", "Q": "GLN",
                           "R": "ARG", "S": "SER", "T": "THR", "V": "VAL", "W": "TRP", "Y": "TYR"}
THREE_TO_ONE_LETTER_MAP = {v: k for k, v in ONE_TO_THREE_LETTER_MAP.items()}
AA_MAP = {aa: i for i, aa in enumerate(sorted(ONE_TO_THREE_LETTER_MAP))}
AA_MAP_INV = {v: k for k, v in AA_MAP.items()}
AA_MAP.update({ONE_TO_THREE_LETTER_MAP[k]: v for k, v in list(AA_MAP.items())})


class ProteinVocabulary(object):
    def __init__(self, add_sos_eos=False):
        self.pad_char, self.unk_char, self.sos_char, self.eos_char = "_", "?", "<", ">"
        self._char2int, self._int2char = {}, {}
        self.stdaas = "".join(AA_MAP_INV[i] for i in range(20))
        for ch in self.stdaas + self.pad_char + self.unk_char + (self.sos_char + self.eos_char if add_sos_eos else ""):
            self.add(ch)
        self.pad_id = self[self.pad_char]
        self.sos_id = self[self.sos_char]
        self.eos_id = self[self.eos_char]

    def __getitem__(self, aa):
        return self._char2int.get(aa, self._char2int[self.unk_char])

    def __contains__(self, aa):
        return aa in self._char2int

    def __setitem__(self, key, value):
        raise ValueError('vocabulary is readonly')

    def __len__(self):
        return len(self._char2int)

    def __repr__(self):
        return f"ProteinVocabulary[size={len(self)}]"

    def int2char(self, id):
        return self._int2char[id]

    def int2chars(self, id):
        return ONE_TO_THREE_LETTER_MAP[self._int2char[id]]

    def add(self, aa):
        if aa in self:
            return self[aa]
        aaid = self._char2int[aa] = len(self._char2int)
        self._int2char[aaid] = aa
        return aaid

    def str2ints(self, seq, add_sos_eos=True):
        body = [self[aa] for aa in seq]
        return [self["<"]] + body + [self[">"]] if add_sos_eos else body

    def ints2str(self, ints, include_sos_eos=False):
        skip = () if include_sos_eos else (self.sos_char, self.eos_char, self.pad_char)
        return "".join(c for c in (self.int2char(i) for i in ints) if c not in skip)


VOCAB = ProteinVocabulary()
