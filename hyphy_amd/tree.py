"""Flat post-order tree arrays — the host-side mirror of ``_TheTree::SetUp``.

Reference: ``src/core/tree.cpp:722-766``.  The reference walks the tree in post-order;
leaves are appended to ``flatLeaves`` and internal nodes to ``flatTree`` (root is the
last internal node).  ``flatParents[k]`` (k < L for leaves, k = L + i for internal node
i) holds the *internal index* of the parent, −1 for the root.  A "node code" n < L is a
leaf, n ≥ L is internal node n − L (``tree_evaluator.cpp:3596-3611``).

Everything here is scalar host work that runs once per partition; none of it is on the
device hot path.
"""
from __future__ import annotations

import dataclasses
from typing import List, Optional, Sequence

import numpy as np


@dataclasses.dataclass
class Node:
    name: str
    children: List["Node"] = dataclasses.field(default_factory=list)
    length: Optional[float] = None
    parent: Optional["Node"] = None

    @property
    def is_leaf(self) -> bool:
        return not self.children


def parse_newick(text: str) -> Node:
    """Minimal Newick reader (names, optional ``:length``; no comments/quotes)."""
    s = text.strip().rstrip(";")
    pos = 0
    auto = [0]

    def parse() -> Node:
        nonlocal pos
        node = Node(name="")
        if s[pos] == "(":
            pos += 1
            while True:
                child = parse()
                child.parent = node
                node.children.append(child)
                if s[pos] == ",":
                    pos += 1
                    continue
                if s[pos] == ")":
                    pos += 1
                    break
                raise ValueError(f"bad newick at {pos}: {s[pos:pos+20]!r}")
        start = pos
        while pos < len(s) and s[pos] not in ",():":
            pos += 1
        node.name = s[start:pos].strip()
        if pos < len(s) and s[pos] == ":":
            pos += 1
            start = pos
            while pos < len(s) and s[pos] not in ",()":
                pos += 1
            node.length = float(s[start:pos])
        if not node.name:
            auto[0] += 1
            node.name = f"Node{auto[0]}"
        return node

    root = parse()
    return root


def to_newick(root: Node, lengths: bool = False) -> str:
    def rec(n: Node) -> str:
        lab = n.name if n.parent is not None else ""
        if lengths and n.length is not None and n.parent is not None:
            lab += f":{n.length:.10g}"
        if n.is_leaf:
            return lab
        return "(" + ",".join(rec(c) for c in n.children) + ")" + lab

    return rec(root)


@dataclasses.dataclass
class FlatTree:
    """Post-order flat arrays (``_TheTree::SetUp``, ``tree.cpp:722-766``)."""

    leaf_names: List[str]          # flatLeaves order
    inode_names: List[str]         # flatTree order, root last
    flat_parents: np.ndarray       # int64[L+I], parent as internal index, root −1
    root: Optional[Node] = None

    @property
    def L(self) -> int:
        return len(self.leaf_names)

    @property
    def I(self) -> int:
        return len(self.inode_names)

    @property
    def n_branches(self) -> int:
        return self.L + self.I - 1

    def branch_names(self) -> List[str]:
        """Name of the branch above node code n (n = 0 … L+I−2; root has none)."""
        return self.leaf_names + self.inode_names[:-1]

    def children_of(self, inode: int) -> List[int]:
        """Node codes of the children of internal node ``inode`` (ascending)."""
        return [int(c) for c in np.nonzero(self.flat_parents == inode)[0]]

    def all_update_nodes(self) -> np.ndarray:
        """Node list of a first evaluation: every branch (``likefunc.cpp:10965-10967``)."""
        return np.arange(self.L + self.I - 1, dtype=np.int64)

    def path_update_nodes(self, node_code: int) -> np.ndarray:
        """Dirty list for a single changed branch ``node_code``, as
        ``DetermineNodesForUpdate`` builds it (``tree.cpp:3117-3331``, esp. 3291-3297):
        the node, all its ancestors (root excluded — it has no branch) and the direct
        children of every touched internal node, ascending."""
        L = self.L
        touched_parents = set()
        out = {int(node_code)}
        n = int(node_code)
        while True:
            p = int(self.flat_parents[n])
            if p < 0:
                break
            touched_parents.add(p)
            if self.flat_parents[L + p] >= 0:
                out.add(L + p)
            n = L + p
        for p in touched_parents:
            out.update(self.children_of(p))
        return np.array(sorted(out), dtype=np.int64)


def flatten(root: Node) -> FlatTree:
    leaves: List[Node] = []
    inodes: List[Node] = []

    # iterative post-order (trees with thousands of taxa would blow the recursion limit)
    stack = [(root, 0)]
    while stack:
        node, idx = stack.pop()
        if idx < len(node.children):
            stack.append((node, idx + 1))
            stack.append((node.children[idx], 0))
        else:
            (leaves if node.is_leaf else inodes).append(node)
    index = {id(n): i for i, n in enumerate(inodes)}
    fp = np.empty(len(leaves) + len(inodes), dtype=np.int64)
    for k, n in enumerate(leaves + inodes):
        fp[k] = -1 if n.parent is None else index[id(n.parent)]
    return FlatTree([n.name for n in leaves], [n.name for n in inodes], fp, root)


def random_tree(n_leaves: int, rng: np.random.Generator, trifurcating_root: bool = True) -> Node:
    """Random topology by random joins; HyPhy trees are unrooted with a trifurcating root
    by default (I = L − 2, B = 2L − 3)."""
    nodes = [Node(name=f"T{k+1}") for k in range(n_leaves)]
    counter = 0
    stop = 3 if (trifurcating_root and n_leaves >= 3) else 2
    while len(nodes) > stop:
        i, j = sorted(rng.choice(len(nodes), size=2, replace=False))
        b = nodes.pop(j)
        a = nodes.pop(i)
        counter += 1
        p = Node(name=f"N{counter}", children=[a, b])
        a.parent = p
        b.parent = p
        nodes.append(p)
    root = Node(name="root", children=nodes)
    for c in nodes:
        c.parent = root
    return root


def caterpillar_tree(n_leaves: int) -> Node:
    """Maximally unbalanced (ladder) tree — deep enough to force underflow rescaling."""
    cur = Node(name="T1")
    for k in range(1, n_leaves - 2):
        leaf = Node(name=f"T{k+1}")
        p = Node(name=f"N{k}", children=[cur, leaf])
        cur.parent = p
        leaf.parent = p
        cur = p
    a = Node(name=f"T{n_leaves-1}")
    b = Node(name=f"T{n_leaves}")
    root = Node(name="root", children=[cur, a, b])
    for c in root.children:
        c.parent = root
    return root


def flat_from_parents(flat_parents: Sequence[int], L: int) -> FlatTree:
    fp = np.asarray(flat_parents, dtype=np.int64)
    I = len(fp) - L
    return FlatTree([f"T{k+1}" for k in range(L)], [f"N{k+1}" for k in range(I)], fp)
