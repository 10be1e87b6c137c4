"""ORACLE SHIM (test infrastructure, this container only): stand-in for `pygtrie.CharTrie`.

Only the three methods the reference uses (/root/reference/pyctcdecode/language_model.py:13,
135,145,183,188,263,331): fromkeys, has_node, iterkeys(prefix, shallow=True).  Children keep
insertion order, like pygtrie's default node storage, so the first key yielded for a prefix is
the first-inserted key below it.
"""


class _Node:
    __slots__ = ("children", "is_key")

    def __init__(self):
        self.children = {}
        self.is_key = False


class CharTrie:
    def __init__(self):
        self._root = _Node()
        self._n = 0

    @classmethod
    def fromkeys(cls, keys, value=None):
        trie = cls()
        for k in keys:
            trie._insert(k)
        return trie

    def _insert(self, key):
        node = self._root
        for ch in key:
            nxt = node.children.get(ch)
            if nxt is None:
                nxt = node.children[ch] = _Node()
            node = nxt
        if not node.is_key:
            node.is_key = True
            self._n += 1

    def _find(self, prefix):
        node = self._root
        for ch in prefix:
            node = node.children.get(ch)
            if node is None:
                return None
        return node

    def has_node(self, prefix):
        node = self._find(prefix)
        if node is None:
            return 0
        if node is self._root and self._n == 0:
            return 0
        return 1 + int(node.is_key)

    def iterkeys(self, prefix="", shallow=False):
        node = self._find(prefix)
        if node is None or (node is self._root and self._n == 0):
            raise KeyError(prefix)

        def walk(n, path):
            if n.is_key:
                yield path
                if shallow:
                    return
            for ch, child in n.children.items():
                yield from walk(child, path + ch)

        return walk(node, prefix)

    def __len__(self):
        return self._n
